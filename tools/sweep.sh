#!/bin/bash
# GPU box: sweep launch geometry / prefetch for the coupled bench; one summary line each
# usage: tools/sweep.sh "<PF list>" "<BPC list>" [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
PFS=${1:-"0 1"}; BPCS=${2:-"2 4 8"}; shift; shift
for PF in $PFS; do for BPC in $BPCS; do
  MSGW_PREFETCH=$PF MSGW_BLOCKS_PER_CU=$BPC timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('PF=$PF BPC=$BPC', 'blocks', d['config']['blocks'], 'value %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kern_ms %.4f' % r['kernel_ms_avg'], 'frac %.3f' % r['frac'])
"
done; done
