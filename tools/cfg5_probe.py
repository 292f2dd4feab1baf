"""GPU box: which source amplitude makes BASELINE config 5 (fp32, online saturation + relaunch) a sane workload?
For each alpha: step time, the wind the waves drive, how many rays saturation has cut and how many have been relaunched."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "python-msgwam_amd"))
import numpy as np
import bench
from msgwam_amd import _capi
from msgwam_amd.spectrum import gaussian_spectrum
n = 1_250_000
lprop, grid, grids, uu, vv = bench.column(101)
for alpha in [float(a) for a in sys.argv[1:]] or [0.02, 0.04, 0.08, 0.5]:
    sp = gaussian_spectrum(n, grids, lprop.rhobar, alpha=alpha)
    p = _capi.Propagator(101, n, dtype="f32")
    p.set_config(0.01, 0.0, 1.0, True)
    p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, uu, vv)
    p.upload_rays(sp["dens"], sp["rr"], sp["drr"], sp["kk"], sp["ll"], sp["mm"], sp["dmm"], sp["phi"], sp["dkk"], sp["dll"], sp["area"])
    out = []
    for k in range(4):
        p.step(120.0, 20, _capi.RELAUNCH); p.sync()
        t0 = time.perf_counter(); p.step(120.0, 200, _capi.RELAUNCH); p.sync(); dt = time.perf_counter() - t0
        dens, rr, mm = p.download_rays()
        u, v = p.download_column()
        live = sp["dens"] > 1e-30 * sp["dens"].max()
        cut = np.mean(dens[live] < 0.999 * sp["dens"][live])
        moved = np.mean(np.abs(rr - sp["rr"]) > 1.0)
        out.append(f"[{220 * (k + 1)} steps: {dt / 200 * 1e6:.1f} us/step, max|u| {np.abs(u).max():.3g}, dens cut {cut:.3f}, away from source {moved:.3f}, z max {rr.max() / 1e3:.0f} km]")
    print(f"alpha {alpha}: " + " ".join(out), flush=True)
    p.close()
