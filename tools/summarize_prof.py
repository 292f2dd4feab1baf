"""Summarise gpurun_out/prof/<tag>/ (kernel trace stats + PMC csv) into profiles/<tag>_summary.json/.md"""
import csv, glob, json, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof", tag)
out = {"tag": tag, "kernels": {}, "pmc": {}}


def newest(pattern):
    """gpurun MERGES a call's output into the local gpurun_out/, so a pass directory can hold the csv files of several
    runs (one per process id): only the most recent one belongs to the current sources."""
    by_dir = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        by_dir[f.split(os.sep + tag + os.sep, 1)[1].split(os.sep)[0]].append(f)
    return [max(fs, key=os.path.getmtime) for fs in by_dir.values()]


for f in newest(os.path.join(src, "trace", "**", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out["kernels"][r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                         min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3,
                                         total_ms=float(r["TotalDurationNs"]) / 1e6, pct=float(r["Percentage"]))
for f in newest(os.path.join(src, "pmc_*", "**", "*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for cn, vals in cs.items():
            out["pmc"].setdefault(k, {})[cn] = dict(mean=sum(vals) / len(vals), n=len(vals))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", f"{tag}_summary.md"), "w") as fh:
    fh.write(f"# rocprofv3 summary `{tag}`\n\n## kernel-trace --stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
        fh.write(f"| `{k[:90]}` | {v['calls']} | {v['avg_us']:.2f} | {v['min_us']:.2f} | {v['max_us']:.2f} | {v['pct']:.1f} |\n")
    fh.write("\n## PMC (mean per dispatch)\n\n")
    for k, cs in out["pmc"].items():
        if "k_ray" not in k and "k_rk3" not in k and "k_column" not in k and "k_flux" not in k:
            continue
        fh.write(f"### `{k[:100]}`\n\n" + "".join(f"- {cn}: {v['mean']:.4g} (n={v['n']})\n" for cn, v in sorted(cs.items())) + "\n")
print(json.dumps({k: v["avg_us"] for k, v in out["kernels"].items() if "msgw" in k}, indent=1))
