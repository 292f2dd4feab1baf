#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + separate PMC passes of ONE command, raw output under
# gpurun_out/prof/<tag>/ (summarise with tools/summarize_prof.py <tag>, which writes profiles/<tag>_summary.{md,json}).
# usage: tools/profile2.sh <tag> <python script relative to the repo root> [args...]     (environment is inherited)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
echo "python3 $SCRIPT $*" > "$OUT/command.txt"
(cd "$ROOT" && python3 -c "import bench; print(bench.kernel_src_digest())") > "$OUT/src_digest.txt" 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$N" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_$N.log" 2>&1 || echo "pmc pass $N failed"
done
tail -1 "$OUT/trace.log" | cut -c1-300
echo "done $TAG"
