set -x
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_c3_driver.json 2> gpurun_out/b_c3_driver.err
python bench.py --no-size-sweep --no-cpu-baseline > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err
python bench.py --workload config5 --no-size-sweep --no-cpu-baseline > gpurun_out/b_c5.json 2> gpurun_out/b_c5.err
MSGW_REGTILES=0 python bench.py --workload config5 --no-size-sweep --no-cpu-baseline > gpurun_out/b_c5_stream.json 2> gpurun_out/b_c5_stream.err
MSGW_PERSIST=0 python bench.py --workload config5 --no-size-sweep --no-cpu-baseline --steps 50 > gpurun_out/b_c5_chain.json 2> gpurun_out/b_c5_chain.err
MSGW_PERSIST=0 python bench.py --no-size-sweep --no-cpu-baseline --steps 50 > gpurun_out/b_c3_chain.json 2> gpurun_out/b_c3_chain.err
python bench.py --workload config2 --no-cpu-baseline --steps 1000 --warmup 100 > gpurun_out/b_c2.json 2> gpurun_out/b_c2.err
python bench.py --workload config5 --rays-per-gpu 1000000 --no-size-sweep --no-cpu-baseline > gpurun_out/b_c5_1e6.json 2> gpurun_out/b_c5_1e6.err
for f in gpurun_out/b_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d['roofline'] or {}
print('  value %.3e  ms/step %.4f [%.4f..%.4f] rep %d  frac %s  kern_ms %s  persist %s res %s finite %s copy %s' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['repeats'], r.get('frac'), r.get('kernel_ms_avg'), d['config']['persist_steps'], d['config']['register_resident_tiles_per_workgroup'], d['state_finite'], r.get('copy_ceiling_gbs')))
"; done
tail -3 gpurun_out/b_*.err
