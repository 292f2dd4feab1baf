"""
Headless counterpart of the reference's experiment script `raytracer.py`
(configuration :32-64, initial condition :71-117, time loop :157-191, wave-action
conservation diagnostics :198-240; the matplotlib part :247-290 is out of scope).

Two ways to run the same experiment on the MI355X path:

  run(mode="dropin")    the reference loop verbatim against the libprop mirror:
                        one `lprop.RK3` + one `lprop.saturation(direct=True)` per
                        step, state crossing PCIe every step (compatibility);
  run(mode="resident")  the state stays in HBM; the post-step saturation (incl. the
                        driver's `/ 1` quirk, raytracer.py:184) is fused into the
                        third RK stage; history rows are downloaded every
                        `snapshot_every` steps (default: every step, as the
                        reference keeps every row).

Both return the same dict of history arrays (`int_*` names as in raytracer.py).
"""
import numpy as np

from . import _capi
from . import libprop as lprop


def configure(ngrid=101, grid_max=100e3, NN=0.01, phi0=0.0, saturate_online=False):
    """raytracer.py:32-64, :74-77, :93-99."""
    lprop.HPROP_GLOBAL = False                                   # :38
    lprop.set_model_setup(bvf=NN, rhs=lprop.rhs_default, boussinesq=False, sig_rr=10000, u0=4,
                          rr0=40000, rr1=40000, phi0=phi0, kappa=1., saturate_online=saturate_online)
    grid = np.linspace(0, grid_max, ngrid)                       # :74
    grids = .5 * (grid[:-1] + grid[1:])                          # :75
    lprop.grid, lprop.grids = grid, grids                        # :76-77
    init_uu = lprop.velocities_sine_homogeneous(grids)           # :93
    init_vv = np.zeros(init_uu.shape)
    lprop.set_hydrostatics()                                     # :98
    lprop.set_pressure_gradient(init_uu, init_vv)                # :99
    return grid, grids, init_uu, init_vv


def initial_rays(nray, grids, alpha=0.01, NN=0.01, phi0=0.0, rr_init_min=0., rr_init_max=15000.):
    """raytracer.py:71-72, :83-92, :102-117."""
    k_abs_init = 2 * np.pi / 50e3
    direction = 90
    ic = {}
    ic["kk"] = np.ones(nray) * k_abs_init * np.sin(np.deg2rad(direction))
    ic["ll"] = np.ones(nray) * k_abs_init * np.cos(np.deg2rad(direction))
    ic["mm"] = np.ones(nray) * -2 * np.pi / 5e3
    ic["lam"] = np.zeros(nray)
    ic["phi"] = np.ones(nray) * phi0
    rr_grid = np.linspace(rr_init_min, rr_init_max, nray + 1)
    ic["rr"] = .5 * (rr_grid[:-1] + rr_grid[1:])
    ic["drr"] = np.ones(nray) * np.diff(ic["rr"])[0]
    rr_mm_area = 5e-5 * ic["drr"]
    ic["dmm"] = rr_mm_area / ic["drr"]
    ic["dll"] = np.ones(nray) * 1e-4
    ic["dkk"] = np.ones(nray) * 1e-4
    lprop.set_statics(dll=ic["dll"], dkk=ic["dkk"], rr_mm_area=rr_mm_area)
    f0 = 2 * lprop.ROT_EARTH * np.sin(phi0)
    rhobar_ray = np.interp(ic["rr"], grids, lprop.rhobar)
    omh_ray = lprop.omega(ic["kk"], ic["ll"], ic["mm"], phi0)
    amplitude = alpha ** 2 * rhobar_ray / 2 * omh_ray / ic["mm"] ** 2 / (omh_ray ** 2 - f0 ** 2) * NN ** 2
    profile = np.exp(-(ic["rr"] - ic["rr"].mean()) ** 2 / 2 / 2000 ** 2)
    ic["dens"] = amplitude * profile / ic["dkk"] / ic["dll"] / ic["dmm"]
    ic["area"] = rr_mm_area
    return ic


KEYS = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]


def run(nray=60, ngrid=101, dt=120, nt_max=1440, alpha=0.01, mode="resident", snapshot_every=1,
        diagnostics=True, ref_quirks=True, progress=False, checkpoint_path=None, checkpoint_every=0, resume_from=None):
    """`checkpoint_path` + `checkpoint_every` (resident mode): the device state is written to that .npz every so many
    steps (at history-row boundaries) and at the end; `resume_from`: continue such a file up to `nt_max` -- the history
    rows before its step are not part of the file and stay zero (`stored` lists the rows that were filled)."""
    grid, grids, init_uu, init_vv = configure(ngrid)
    ic = initial_rays(nray, grids, alpha)
    saturate_online = lprop.model_config['saturate_online']
    H = {f"int_{k}": np.zeros((nt_max + 1, nray)) for k in KEYS}
    H["int_dens_prop"] = np.zeros((nt_max + 1, nray))
    H["int_uu"] = np.zeros((nt_max + 1, len(grids)))
    H["int_vv"] = np.zeros((nt_max + 1, len(grids)))
    for k in KEYS:
        H[f"int_{k}"][0] = ic[k]
    H["int_dens_prop"][0] = ic["dens"]
    H["int_uu"][0], H["int_vv"][0] = init_uu, init_vv
    stored = [0]

    if mode == "dropin":                                         # raytracer.py:157-191
        for nt in range(1, nt_max + 1):
            state_in = np.empty(11, dtype=object)
            for i, k in enumerate(KEYS):
                state_in[i] = H[f"int_{k}"][nt - 1]
            state_in[9], state_in[10] = H["int_uu"][nt - 1], H["int_vv"][nt - 1]
            state_out = lprop.RK3(dt, state_in)                  # :175
            H["int_dens_prop"][nt] = state_out[0]
            for i, k in enumerate(KEYS):
                if k != "dens":
                    H[f"int_{k}"][nt] = state_out[i]
            H["int_uu"][nt], H["int_vv"][nt] = state_out[9], state_out[10]
            if not saturate_online:                              # :182-188
                div = 1 if ref_quirks else dt
                H["int_dens"][nt] = lprop.saturation(
                    dt, H["int_dens_prop"][nt], H["int_rr"][nt - 1], (H["int_rr"][nt] - H["int_rr"][nt - 1]) / div,
                    H["int_drr"][nt - 1], (H["int_drr"][nt] - H["int_drr"][nt - 1]) / dt,
                    H["int_kk"][nt], H["int_ll"][nt], H["int_mm"][nt - 1],
                    (H["int_mm"][nt] - H["int_mm"][nt - 1]) / dt, direct=True)
            else:
                H["int_dens"][nt] = H["int_dens_prop"][nt]
            stored.append(nt)
            if progress:
                print('progress: {0:.2f}%'.format(nt / nt_max * 100), end='\r')
        lprop.release_device()
    elif mode == "resident":
        nt = 0
        if resume_from is not None:
            p, meta = _capi.Propagator.load_checkpoint(resume_from)
            nt = int(meta["step"])
            if p.n != nray or p.ngrid != ngrid or float(meta["dt"]) != float(dt):
                p.close()
                raise ValueError("the checkpoint belongs to another experiment (nray, ngrid or dt differ)")
            stored = []
        else:
            p = _capi.Propagator(ngrid, nray)
            p.set_config(lprop.model_config['bvf'], lprop.model_config['phi0'], lprop.model_config['kappa'],
                         saturate_online)
            p.set_column(grid, grids, lprop.rhobar, lprop.pressure_gradient, init_uu, init_vv)
            p.upload_rays(ic["dens"], ic["rr"], ic["drr"], ic["kk"], ic["ll"], ic["mm"], ic["dmm"], ic["phi"],
                          ic["dkk"], ic["dll"], ic["area"])
        flags = 0
        if not saturate_online:
            flags = _capi.DIRECT_SAT_QUIRK if ref_quirks else _capi.DIRECT_SAT
        last_ckpt = nt
        while nt < nt_max:
            k = min(snapshot_every, nt_max - nt)
            p.step(dt, k, flags)
            nt += k
            dens, rr, mm = p.download_rays()
            uu, vv = p.download_column()
            H["int_dens"][nt], H["int_rr"][nt], H["int_mm"][nt] = dens, rr, mm
            H["int_dens_prop"][nt] = dens       # only the saturated density is kept resident
            for kname in ("lam", "phi", "drr", "kk", "ll", "dmm"):
                H[f"int_{kname}"][nt] = ic[kname]
            H["int_uu"][nt], H["int_vv"][nt] = uu, vv
            stored.append(nt)
            if checkpoint_path and checkpoint_every and (nt - last_ckpt >= checkpoint_every or nt == nt_max):
                p.save_checkpoint(checkpoint_path, step=nt, dt=dt)
                last_ckpt = nt
        p.close()
    else:
        raise ValueError("mode must be 'dropin' or 'resident'")
    H["stored"] = np.array(stored)

    if diagnostics:                                              # raytracer.py:198-240 on the stored rows
        rows = H["stored"]
        wa = np.zeros((len(rows), len(grids)))
        flux_diag = np.zeros((len(rows), len(grids) - 1))
        for i, nt in enumerate(rows):
            rr_down = H["int_rr"][nt] - .5 * H["int_drr"][nt]
            rr_up = H["int_rr"][nt] + .5 * H["int_drr"][nt]
            mm_down = H["int_mm"][nt] - .5 * H["int_dmm"][nt]
            mm_up = H["int_mm"][nt] + .5 * H["int_dmm"][nt]
            wa[i] = lprop.wave_projection(H["int_dens"][nt], H["int_lam"][nt], H["int_phi"][nt], rr_down, rr_up,
                                          H["int_kk"][nt], H["int_ll"][nt], mm_down, mm_up,
                                          ic["dkk"], ic["dll"], H["int_dmm"][nt], grid, var=2)        # :213
            flux_diag[i] = lprop.wave_projection(H["int_dens"][nt], H["int_lam"][nt], H["int_phi"][nt], rr_down,
                                                 rr_up, H["int_kk"][nt], H["int_ll"][nt], mm_down, mm_up,
                                                 ic["dkk"], ic["dll"], H["int_dmm"][nt], grids, var=1)  # :227
        dz = np.diff(grid[:2])[0]
        prop_diag = np.zeros((len(rows), len(grids)))
        prop_diag[:, 1:-1] = -np.diff(flux_diag, axis=-1) / dz                                          # :237
        H.update(wa=wa, flux_diag=flux_diag, prop_diag=prop_diag)
        lprop.release_device()
    return H


if __name__ == "__main__":
    import argparse
    import time
    ap = argparse.ArgumentParser(description="headless raytracer.py on the MI355X path")
    ap.add_argument("--nray", type=int, default=60)
    ap.add_argument("--steps", type=int, default=1440)
    ap.add_argument("--mode", default="resident", choices=["resident", "dropin"])
    ap.add_argument("--snapshot-every", type=int, default=1)
    a = ap.parse_args()
    t0 = time.perf_counter()
    out = run(nray=a.nray, nt_max=a.steps, mode=a.mode, snapshot_every=a.snapshot_every)
    print(f"{a.mode}: {a.nray} rays x {a.steps} steps in {time.perf_counter() - t0:.2f} s; "
          f"int_rr sum {out['int_rr'].sum():.9e}  wa sum {out['wa'].sum():.9e}")
