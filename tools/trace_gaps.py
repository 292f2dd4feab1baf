"""Read a rocprofv3 kernel_trace.csv: per-kernel average duration and the idle gap before each kernel."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void msgw::", "").replace("msgw::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if prev_end is not None:
        gap[n].append((s - prev_end) / 1e3)
    prev_end = e
tot = 0
for n in dur:
    d = sorted(dur[n]); g = sorted(gap[n]) if gap[n] else [0]
    print(f"{n[:60]:60s} calls {len(d):5d} dur med {d[len(d)//2]:7.2f} us  gap-before med {g[len(g)//2]:6.2f} us")
