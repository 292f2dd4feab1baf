#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + PMC passes of bench.py, raw output under gpurun_out/prof/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# warmup == steps, no size sweep, separate event pass: three launches of the persistent kernel, each covering the same 30 steps, so the
# --stats average is the duration bench.py reports as roofline.kernel_ms_avg for this very command
ARGS="--steps 30 --warmup 30 --no-cpu-baseline --no-size-sweep --kernel-events separate $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $N failed"
done
echo done
