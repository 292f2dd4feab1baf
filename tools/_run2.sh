one() { env $1 python bench.py --steps 100 --warmup 20 --no-size-sweep --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2: %.2f us/step  value %.3e res %s frac %.3f' % (d['ms_per_step']*1e3, d['value'], d['config']['register_resident_tiles_per_workgroup'], d['roofline']['frac']))"; }
one A=1 "--workload config3"
one A=1 "--workload config3 --rays-per-gpu 2000000"
one A=1 "--workload config3 --rays-per-gpu 3000000"
one A=1 "--workload config3 --rays-per-gpu 4000000"
one A=1 "--workload config3 --rays-per-gpu 8000000"
one MSGW_REGTILES=0 "--workload config3 --rays-per-gpu 8000000"
one MSGW_REGTILES=0 "--workload config3 --rays-per-gpu 4000000"
