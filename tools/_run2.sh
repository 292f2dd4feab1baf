one() { (cd $1 && env $3 python bench.py --steps 200 --warmup 20 --no-size-sweep --no-cpu-baseline $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 $4: %.2f us/step  value %.3e' % (d['ms_per_step']*1e3, d['value']))"); }
one . x A=1 "--workload config5"
one . x MSGW_REGTILES=0 "--workload config5"
one . x A=1
one . x MSGW_REGTILES=0
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
