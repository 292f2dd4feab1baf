"""GPU box: what does the (height bin, group-velocity bin) order give for the bench workload?"""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
lprop, grid, grids, uu, vv = bench.column(101)
n = 1_000_000
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
def span_of(rr, W=128):
    lo, up = rr - 75.0, rr + 75.0
    nl = np.clip(np.trunc(lo / 1000.0), 0, 98).astype(int); nu = np.clip(np.trunc(up / 1000.0 + 1), 0, 98).astype(int)
    m = (len(rr) // W) * W
    wl = nl[:m].reshape(-1, W).min(1); wu = nu[:m].reshape(-1, W).max(1)
    s = np.where(wu > wl, wu - wl, 0)
    return round(float(s.mean()), 2), round(float(np.mean(s > 8)), 3)
def cg_of(kk, ll, mm, bvf=0.01):
    kh2 = kk * kk + ll * ll; vk2 = kh2 + mm * mm
    om = np.sqrt(bvf ** 2 * kh2 / vk2)
    return -mm * om * om / om / vk2
p = _capi.Propagator(101, n)
p.set_config(0.01, 0.0, 1.0, False)
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
for upto in (0, 64, 192):
    p.step(120.0, upto - (0 if upto == 0 else prev)) if upto else None
    prev = upto
    _, rr, mm = p.download_rays()
    cg = cg_of(sp["kk"], sp["ll"], mm)
    c0 = np.mean(np.abs(cg))
    t = cg / (np.abs(cg) + c0)
    res = {"natural": span_of(rr)}
    for zdiv, ncg in ((4, 256), (4, 64), (1, 256), (16, 256)):
        zb = np.clip(np.floor(rr / (1000.0 / zdiv)) + 1, 0, 100 * zdiv + 1).astype(np.int64)
        cb = np.clip(((t + 1) * 0.5 * ncg).astype(np.int64), 0, ncg - 1)
        order = np.argsort(zb * ncg + cb, kind="stable")
        res[f"z/{zdiv} cg{ncg}"] = span_of(rr[order])
    order = np.argsort(rr, kind="stable"); res["by height"] = span_of(rr[order])
    print(f"step {upto}: (mean span, frac>8) {res}")
    # how fast does each order spread?  advance 64 more steps in natural order and look at the same groupings
p.close()
