"""GPU box, diagnostic build (tools/_build/libmsgwam_hip_stamp.so, -DMSGW_STAMP): per-workgroup,
per-pass timeline of the persistent kernel for the bench workload."""
import ctypes as C, os, sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["MSGW_LIBRARY"] = os.path.join(R, "tools", "_build", "libmsgwam_hip_stamp.so")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lprop, grid, grids, uu, vv = bench.column(101)
# MSGW_TL="f32,sat,rl,alpha=0.5": the config-5 kernel variant instead of the default float64 plain one
opts = os.environ.get("MSGW_TL", "").split(",")
alpha = float(([o[6:] for o in opts if o.startswith("alpha=")] or ["0.01"])[0])
flags = _capi.RELAUNCH if "rl" in opts else 0
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=alpha)
p = _capi.Propagator(101, n, dtype="f32" if "f32" in opts else "f64")
p.set_config(0.01, 0.0, 1.0, "sat" in opts)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
p.set_tuning(int(os.environ.get("MSGW_BLOCKS_PER_CU", 3)), 0)
p.step(120.0, int(os.environ.get("MSGW_TL_PRE", 20)), flags); p.sync(); print("persist_steps", p.counters()["persist_steps"], "blocks", p.counters()["blocks"])
p.step(120.0, 5, flags); p.sync()
nb = p.counters()["blocks"]; NP = 16
buf = np.zeros((nb, NP, 4), dtype=np.uint64)
rc = p.lib.msgw_debug_stamps(p.ctx, buf.ctypes.data_as(C.c_void_p), nb)
assert rc == 0
buf = buf[buf[:, 0, 0] > 0]                         # service workgroups record nothing
t0 = buf[:, 0, 0].min()
T = (buf.astype(np.float64) - float(t0)) * 0.01     # us
names = ["pass start", "column done", "tiles done", "published"]
for q in range(0, 10):
    line = f"pass {q:2d}: "
    for k in range(4):
        v = T[:, q, k]
        line += f"{names[k]} [{v.min():7.1f} {np.median(v):7.1f} {v.max():7.1f}]  "
    print(line)
d_wait = np.median(T[:, 3:9, 1] - T[:, 3:9, 0]); d_tiles = np.median(T[:, 3:9, 2] - T[:, 3:9, 1]); d_pub = np.median(T[:, 3:9, 3] - T[:, 3:9, 2])
print(f"median per pass: wait+column {d_wait:.2f} us, tiles {d_tiles:.2f} us, publish {d_pub:.2f} us; pass period {(np.median(T[:, 9, 0]) - np.median(T[:, 3, 0])) / 6:.2f} us")
# per-workgroup detail of one steady pass: by dispatch round (workgroup index // 256)
q = 6
for rnd in range((nb + 255) // 256):
    sel = np.arange(len(T)) // 256 == rnd
    if not sel.any():
        continue
    base = np.median(T[:, q, 0])
    print(f"round {rnd}: start {np.median(T[sel, q, 0]) - base:6.2f}  column {np.median(T[sel, q, 1] - T[sel, q, 0]):5.2f}  "
          f"tiles {np.median(T[sel, q, 2] - T[sel, q, 1]):5.2f}  publish {np.median(T[sel, q, 3] - T[sel, q, 2]):5.2f}  "
          f"end {np.median(T[sel, q, 3]) - base:6.2f}")
# idle estimate: gap between a workgroup's publish end and its next pass start, and wait inside column
print("gap publish->next start (median)", np.median(T[:, 4:9, 0] - T[:, 3:8, 3]))
# which workgroups are the stragglers?
q = 6
dur = T[:, q, 2] - T[:, q, 1]
order = np.argsort(-T[:, q, 0])
print("latest starters of pass", q, ":", [(int(i), round(float(T[i, q, 0] - np.median(T[:, q, 0])), 1), round(float(dur[i]), 1)) for i in order[:24]])
late = np.sort(order[:48])
print("late set indices:", late.tolist())
print("tiles duration by index decile:", [round(float(np.median(dur[k * len(dur) // 10:(k + 1) * len(dur) // 10])), 1) for k in range(10)])
pub = T[:, q, 3] - T[:, q, 2]
print("publish > 4us:", np.nonzero(pub > 4)[0].tolist()[:64])
hw = buf[:, NP - 1, 0].astype(np.int64); xcc = buf[:, NP - 1, 1].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
print("xcc of workgroups 0..15:", xcc[:16].tolist())
for x in range(8):
    sel = xcc == x
    print(f"xcc {x}: workgroups {int(sel.sum())}, distinct CUs {len(np.unique(cuid[sel]))}, max per CU {np.bincount(np.unique(cuid[sel], return_inverse=True)[1]).max()}, "
          f"tiles median {np.median(dur[sel]):.1f} us, start offset {np.median(T[sel, q, 0]) - np.median(T[:, q, 0]):.1f} us")
# breakdown of the trailing cohort (latest 5 % starters of pass q) vs everybody
late = order[:max(len(order) // 20, 8)]
for name, sel in (("trailing 5%", late), ("all", np.arange(len(T)))):
    print(f"{name:12s}: start {np.median(T[sel, q, 0]) - np.median(T[:, q, 0]):6.2f}  wait+column {np.median(T[sel, q, 1] - T[sel, q, 0]):5.2f}  "
          f"tiles {np.median(T[sel, q, 2] - T[sel, q, 1]):5.2f}  publish {np.median(T[sel, q, 3] - T[sel, q, 2]):5.2f}  "
          f"period {np.median(T[sel, q + 1, 0] - T[sel, q, 0]):5.2f}  xcc {np.bincount(xcc[sel], minlength=8).tolist()}  round {np.bincount(sel // 256, minlength=4).tolist()}")
print("publish max", float(pub.max()), "at", int(pub.argmax()), " publish > 4us count", int((pub > 4).sum()))
# steady-state averages over passes 8..14
qs = np.arange(8, 15)
med0 = np.median(T[:, qs, 0], axis=0)
off = (T[:, qs, 0] - med0).mean(axis=1)
wc = (T[:, qs, 1] - T[:, qs, 0]).mean(axis=1); tl = (T[:, qs, 2] - T[:, qs, 1]).mean(axis=1); pb = (T[:, qs, 3] - T[:, qs, 2]).mean(axis=1)
per = (T[:, 14, 0] - T[:, 8, 0]) / 6
o2 = np.argsort(-off)
for name, sel in (("trailing 5%", o2[:len(o2) // 20]), ("next 20%", o2[len(o2) // 20:len(o2) // 4]), ("leading 25%", o2[-len(o2) // 4:]), ("all", o2)):
    print(f"steady {name:12s}: start {off[sel].mean():6.2f}  wait+column {wc[sel].mean():5.2f}  tiles {tl[sel].mean():5.2f}  publish {pb[sel].mean():5.2f}  "
          f"period {per[sel].mean():5.2f}  round {np.bincount(sel // 256, minlength=4).tolist()}  xcc {np.bincount(xcc[sel], minlength=8).tolist()}")
# release time of each pass = earliest 'column done' ; trailing end = latest 'published'
rel = T[:, 6:15, 1].min(axis=0); endmax = T[:, 4:13, 3].max(axis=0)
print("release(q) - max published(q-2):", np.round(rel - endmax, 2).tolist())
print("release period:", np.round(np.diff(rel), 2).tolist())
