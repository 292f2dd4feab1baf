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
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
p = _capi.Propagator(101, n)
p.set_config(0.01, 0.0, 1.0, False)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
p.set_tuning(int(os.environ.get("MSGW_BLOCKS_PER_CU", 3)), 0)
p.step(120.0, 20); p.sync(); print("persist_steps", p.counters()["persist_steps"], "blocks", p.counters()["blocks"])
p.step(120.0, 5); p.sync()
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
