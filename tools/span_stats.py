"""GPU box: per-wavefront level span of the deposit for the bench workload after k steps."""
import sys, os
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
lprop, grid, grids, uu, vv = bench.column(101)
n = 1_000_000
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
p = _capi.Propagator(101, n)
p.set_config(0.01, 0.0, 1.0, False)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
done = 0
for k in (0, 30, 130, 230, 530, 1030):
    p.step(120.0, k - done); done = k
    _, rr, mm = p.download_rays()
    lo, up = rr - 75.0, rr + 75.0
    nl = np.clip(np.trunc(lo / 1000.0), 0, 98).astype(int); nu = np.clip(np.trunc(up / 1000.0 + 1), 0, 98).astype(int)
    ood = ((nl >= 98) & (nu >= 98)) | ((np.trunc(lo/1000.0) <= 0) & (np.trunc(up/1000.0+1) <= 0))
    nl[ood] = 10**6; nu[ood] = -1
    W = 128
    m = (n // W) * W
    wl = nl[:m].reshape(-1, W).min(1); wu = nu[:m].reshape(-1, W).max(1)
    span = np.where(wu > wl, wu - wl, 0)
    print(f"step {k:5d}: mean span/wave {span.mean():.2f}  max {span.max()}  frac>8 {np.mean(span>8):.3f}  rr range [{rr.min():.0f}, {rr.max():.0f}]  active waves {np.mean(span>0):.3f}")
