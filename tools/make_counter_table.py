"""Build profiles/traffic.json -- what bench.py's `roofline.traffic` / `roofline.achieved` (valu_issue) read -- from the
rocprofv3 PMC summaries under profiles/ (tools/profile2.sh -> tools/summarize_prof.py).

    python tools/make_counter_table.py r03_config3=config3:f64:1000000:res4:30 r03_config2=config2:f64:100000:res0:narrow:1000 ...

Each argument is <summary tag>=<table key>:<RK3 steps of one profiled launch>.  Per kernel flavour (the dominant msgw
kernel of the summary) it records, per ray and RK3 step of that launch,
  bytes_per_ray_step            (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / (rays * steps): HBM bytes by the counters; FETCH_SIZE is
                                doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half)
  bytes_per_ray_launch          the same per launch, for the fused fixed-background kernel (state read and written once per launch)
  valu_wave_insts_per_ray_step  SQ_INSTS_VALU / (rays * steps): wave64 VALU instructions
  src_digest                    digest of python-msgwam_amd/csrc at the time of the passes (gpurun_out/prof/<tag>/src_digest.txt)
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
path = os.path.join(ROOT, "profiles", "traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table["_note"] = ("per kernel flavour, from committed rocprofv3 --pmc passes (tools/profile2.sh, separate passes; "
                  "tools/make_counter_table.py): HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md) and wave64 VALU instructions = SQ_INSTS_VALU, per ray and RK3 step of the profiled "
                  "launch.  Keys: workload:dtype:rays_per_gpu:resN[:narrow] (N = register-resident tiles per workgroup; narrow "
                  "= one ray per lane).  src_digest = bench.kernel_src_digest() at the time of the passes; bench.py marks an "
                  "entry stale when the kernel sources have changed since.")
for arg in sys.argv[1:]:
    tag, spec = arg.split("=")
    *key, steps = spec.split(":")
    key, steps = ":".join(key), int(steps)
    rays = int(key.split(":")[2])
    summ = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_summary.json")))
    name = max((k for k in summ["kernels"] if "msgw::k_r" in k), key=lambda k: summ["kernels"][k]["total_ms"])
    pmc = summ["pmc"][name]
    e = {"source": f"profiles/{tag}_summary.md", "kernel": name[:80], "kernel_avg_us": summ["kernels"][name]["avg_us"]}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        b = (2 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024
        if key.startswith("config2"):
            e["bytes_per_ray_launch"] = b / rays
        else:
            e["bytes_per_ray_step"] = b / (rays * steps)
    if "SQ_INSTS_VALU" in pmc:
        e["valu_wave_insts_per_ray_step"] = pmc["SQ_INSTS_VALU"]["mean"] / (rays * steps)
    if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        e["valu_busy"] = 4 * pmc["SQ_ACTIVE_INST_VALU"]["mean"] / 1024 / (pmc["GRBM_GUI_ACTIVE"]["mean"] / 8)
    d = os.path.join(ROOT, "gpurun_out", "prof", tag, "src_digest.txt")
    if os.path.exists(d):
        e["src_digest"] = open(d).read().strip()
    table[key] = e
    print(key, json.dumps(e))
json.dump(table, open(path, "w"), indent=1)
