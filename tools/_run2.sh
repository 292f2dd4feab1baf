one() { env $1 python bench.py --steps 200 --warmup 20 --no-size-sweep --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2: %.2f us/step  value %.3e' % (d['ms_per_step']*1e3, d['value']))"; }
one A=1 "--workload config3"
one MSGW_WHOLE_TILES=1 "--workload config3"
one A=1 "--workload config5"
one MSGW_WHOLE_TILES=1 "--workload config5"
one A=1 "--workload config3 --rays-per-gpu 1200000"
one MSGW_WHOLE_TILES=1 "--workload config3 --rays-per-gpu 1200000"
one MSGW_REGTILES=0 "--workload config3"
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
