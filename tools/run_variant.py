"""GPU box: run ONE kernel path for a number of steps (target of rocprofv3; bench.py covers the persistent and the
fused fixed-background kernels).  usage: run_variant.py KIND [rays] [steps], KIND = tall301 | tall201 | a '_'-joined set
of hprop, nz, sat, direct, rl, f32 (e.g. hprop, nz_sat, hprop_nz, hprop_f32, hprop_direct_rl): the general chain kernel.
The bytes are the library's own SURVEY-8d accounting (counters.algorithmic_bytes_total)."""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import time
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
kind = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ngrid = int(kind[4:]) if kind.startswith("tall") else 101
opts = set() if kind.startswith("tall") else set(kind.split("_"))
assert opts <= {"hprop", "nz", "sat", "direct", "rl", "f32"}, kind
lprop, grid, grids, uu, vv = bench.column(ngrid)
sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=0.01)
rng = np.random.default_rng(0)
p = _capi.Propagator(ngrid, n, dtype="f32" if "f32" in opts else "f64")
phi = rng.uniform(-0.5, 0.5, n) if "hprop" in opts else sp["phi"]
p.set_config(0.01, 0.4 if "hprop" in opts else 0.0, 1.0, "sat" in opts, hprop="hprop" in opts)
if "nz" in opts:
    p.set_bvf_column(0.01 * (1 + 0.2 * grids / grids[-1]))
p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], phi, sp["dkk"], sp["dll"], sp["area"])
if "hprop" in opts:
    p.upload_hprop(np.zeros(n), phi)
flags = (_capi.DIRECT_SAT if "direct" in opts else 0) | (_capi.RELAUNCH if "rl" in opts else 0)
p.step(120.0, 10, flags); p.sync()
b0 = p.counters()["algorithmic_bytes_total"]
t0 = time.perf_counter(); p.step(120.0, steps, flags); p.sync(); dt = time.perf_counter() - t0
nbytes = p.counters()["algorithmic_bytes_total"] - b0
print(f"{kind}: {dt / steps * 1e6:.1f} us/step  {n * steps / dt:.3e} ray-steps/s  {nbytes / dt / 1e9:.0f} GB/s = "
      f"{nbytes / dt / 8e12:.2f} of 8 TB/s for {nbytes / n / steps:.0f} B per ray-step")
p.close()
