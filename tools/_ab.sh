set -e
mkdir -p gpurun_out
B="python bench.py --no-size-sweep --no-cpu-baseline"
X=$PWD/python-msgwam_amd/msgwam_amd/libmsgwam_exp3.so
for n in 768000 1000000 1152000; do
$B --rays-per-gpu $n > gpurun_out/ab_ref_$n.json 2>&1
MSGW_REGTILES=2 $B --rays-per-gpu $n > gpurun_out/ab_ref2t_$n.json 2>&1
MSGW_LIBRARY=$X MSGW_REGTILES=2 $B --rays-per-gpu $n > gpurun_out/ab_exp3_$n.json 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['ms_per_step']*1e3,2), 'us', '%.3e'%d['value'], d['config'].get('blocks'), d['config'].get('register_resident_tiles_per_workgroup'))
    except Exception as e: print(f, 'ERR', e)
PY
