"""Condense hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin) into one line per kernel:
demangled name, VGPRs, SGPRs, spills, scratch bytes per lane, occupancy, static LDS."""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                       capture_output=True, text=True).stdout.splitlines()
print("# kernel | VGPRs | SGPRs | VGPR spill | SGPR spill | scratch B/lane | waves/SIMD | static LDS B")
for r, n in zip(rows, names):
    n = re.sub(r"^void msgw::", "", n)
    n = re.sub(r"\(.*\)$", "", n)
    print(" | ".join([n, r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("VGPRs Spill", "?"),
                      r.get("SGPRs Spill", "?"), r.get("ScratchSize", "?"), r.get("Occupancy", "?"),
                      r.get("LDS Size", "?")]))
