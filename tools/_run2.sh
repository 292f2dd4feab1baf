one() { env $1 python bench.py --steps 200 --warmup 20 --no-size-sweep --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2: %.2f us/step  value %.3e res %s' % (d['ms_per_step']*1e3, d['value'], d['config']['register_resident_tiles_per_workgroup']))"; }
one MSGW_REGTILES=2 "--workload config3"
one MSGW_REGTILES=4 "--workload config3"
one MSGW_REGTILES=4 "--workload config3 --rays-per-gpu 2000000"
one MSGW_REGTILES=4 "--workload config3 --rays-per-gpu 4000000"
python tools/variant_bench.py 1250000 f32 0.01 2>&1 | grep -E "plain|relaunch  |latitude  "
python tools/variant_bench.py 1000000 f64 0.01 2>&1 | grep -E "plain|relaunch  |latitude  "
