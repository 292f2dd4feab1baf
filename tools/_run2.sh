one() { (cd $1 && env $3 python bench.py --steps 200 --warmup 20 --no-size-sweep --no-cpu-baseline $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 $4: %.2f us/step [%s %s]  value %.3e frac %.3f' % (d['ms_per_step']*1e3, d.get('ms_per_step_min'), d.get('ms_per_step_max'), d['value'], d['roofline']['frac']))"); }
L=$PWD/tools/_build/abl/libabl1.so
one . x A=1 "--workload config5"
one . x MSGW_LIBRARY=$L "--workload config5"
one . x MSGW_REGTILES=0 "--workload config5"
one . x "MSGW_REGTILES=0 MSGW_LIBRARY=$L" "--workload config5"
one . x A=1
one . x MSGW_LIBRARY=$L
