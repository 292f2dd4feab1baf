ROOT=$PWD
mkdir -p $ROOT/gpurun_out/pmcab; cd /tmp && export TMPDIR=/tmp
for tree in tools/_build/r1 .; do
  tag=$(echo $tree | tr '/.' '__')
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_FLAT"; do
    N=$(echo $C | cut -c1-20 | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/pmcab/$tag/$N -- python3 $ROOT/$tree/bench.py --steps 30 --warmup 30 --no-cpu-baseline --no-size-sweep --kernel-events none > $ROOT/gpurun_out/pmcab/$tag.$N.log 2>&1 || echo fail $tree
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmcab'
for tag in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, tag)): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, tag, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_rk3_persist' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    print(tag, {k: '%.4g' % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
