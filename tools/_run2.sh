cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for A in 0.01 0.5; do
  O=$R/gpurun_out/prof/f32_a$A; mkdir -p $O
  python3 $R/tools/run_f32.py $A
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-30)
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -- python3 $R/tools/run_f32.py $A > $O/pmc_$N.log 2>&1 || echo fail
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for A in ("0.01", "0.5"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/prof/f32_a{A}/pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rk3_persist" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("alpha", A, {k: "%.4g" % (v[-1]) for k, v in sorted(acc.items())})
PY
