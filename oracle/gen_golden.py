"""
Generate the golden fixtures in tests/golden/ by RUNNING THE REAL REFERENCE
(`/root/reference/lib/libprop.py`, imported read-only, no bytecode written).

Runs only in the build container (the reference never travels to the GPU box);
the resulting small .npz files are committed.  Each file stores the inputs AND
the reference outputs, so tests need nothing but numpy to use them.

    python -B oracle/gen_golden.py            # rewrites tests/golden/*.npz

Cases:
  G1  single rhs_default evaluation, 257 random rays, f=0 / phi0=45deg,
      saturate_online on/off  -> 11 tendencies
  G2  wave_projection var 0,1,2 on grid and grids: edge-case table + 1000 rays
  G3  RK3: coupled driver regime 1/10/100 steps (60 rays, ngrid 101);
      fixed background 1/10/100/1000 steps (100 rays, ngrid 201 = config 1)
  G4  saturation: online (777 rays, strong amplitude, 60 steps); direct = the
      driver's own loop incl. the `/1` quirk, rows 1,10,100,710,1000,1440
  G5  synthetic Gaussian spectrum (the bench workload) 2000 rays, 3 coupled steps
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MSGWAM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402
import lib.libprop as lprop  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def obj_state(arrs):
    """Build the 11-slot object array without numpy collapsing it to 2-D."""
    st = np.empty(len(arrs), dtype=object)
    for i, a in enumerate(arrs):
        st[i] = np.asarray(a, dtype=np.float64)
    return st


def configure(ngrid=101, grid_max=100e3, phi0=0.0, kappa=1.0, saturate_online=False,
              u0=4.0, rr0=40000.0, sig_rr=10000.0, bvf=0.01, rhs=None):
    lprop.HPROP_GLOBAL = False                      # raytracer.py:38
    lprop.set_model_setup(bvf=bvf, rhs=rhs or lprop.rhs_default, boussinesq=False,
                          sig_rr=sig_rr, u0=u0, rr0=rr0, rr1=40000, phi0=phi0,
                          kappa=kappa, saturate_online=saturate_online,
                          hh=8500, rhobar0=1.2)
    grid = np.linspace(0, grid_max, ngrid)
    grids = .5 * (grid[:-1] + grid[1:])
    lprop.grid = grid
    lprop.grids = grids
    uu = lprop.velocities_sine_homogeneous(grids)
    vv = np.zeros(uu.shape)
    lprop.set_hydrostatics()
    lprop.set_pressure_gradient(uu, vv)
    return grid, grids, uu, vv


def driver_ic(nray, grids, alpha=0.01, phi0=0.0, NN=0.01):
    """raytracer.py:71-117 initial condition."""
    k_abs = 2 * np.pi / 50e3
    direction = 90
    kk = np.ones(nray) * k_abs * np.sin(np.deg2rad(direction))
    ll = np.ones(nray) * k_abs * np.cos(np.deg2rad(direction))
    mm = np.ones(nray) * -2 * np.pi / 5e3
    lam = np.zeros(nray)
    phi = np.ones(nray) * phi0
    rr_grid = np.linspace(0, 15000, nray + 1)
    rr = .5 * (rr_grid[:-1] + rr_grid[1:])
    drr = np.ones(nray) * np.diff(rr)[0]
    area = 5e-5 * drr
    dmm = area / drr
    dll = np.ones(nray) * 1e-4
    dkk = np.ones(nray) * 1e-4
    lprop.set_statics(dll=dll, dkk=dkk, rr_mm_area=area)
    f0 = 2 * lprop.ROT_EARTH * np.sin(phi0)
    rhobar_ray = np.interp(rr, grids, lprop.rhobar)
    omh = lprop.omega(kk, ll, mm, phi0)
    amp = alpha ** 2 * rhobar_ray / 2 * omh / mm ** 2 / (omh ** 2 - f0 ** 2) * NN ** 2
    profile = np.exp(-(rr - rr.mean()) ** 2 / 2 / 2000 ** 2)
    dens = amp * profile / dkk / dll / dmm
    return dict(dens=dens, lam=lam, phi=phi, rr=rr, drr=drr, kk=kk, ll=ll, mm=mm,
                dmm=dmm, dkk=dkk, dll=dll, area=area)


def random_rays(rng, n, grids, phi0, amp_scale=1.0, zlo=-2e3, zhi=110e3):
    rr = rng.uniform(zlo, zhi, n)
    drr = rng.uniform(100., 3000., n)
    kh = 2 * np.pi / rng.uniform(20e3, 200e3, n)
    az = rng.uniform(0, 2 * np.pi, n)
    kk = kh * np.sin(az)
    ll = kh * np.cos(az)
    mm = -2 * np.pi / rng.uniform(1e3, 20e3, n) * rng.choice([-1.0, 1.0], n)
    area = 5e-5 * drr * rng.uniform(0.5, 2.0, n)
    dmm = area / drr
    dkk = np.ones(n) * 1e-4
    dll = np.ones(n) * 1e-4
    lam = np.zeros(n)
    phi = np.ones(n) * phi0
    lprop.set_statics(dll=dll, dkk=dkk, rr_mm_area=area)
    f0 = 2 * lprop.ROT_EARTH * np.sin(phi0)
    rhobar_ray = np.interp(rr, grids, lprop.rhobar)
    omh = lprop.omega(kk, ll, mm, phi0)
    amp = rhobar_ray / 2 * omh / mm ** 2 / (omh ** 2 - f0 ** 2) * 0.01 ** 2
    dens = amp_scale * rng.uniform(0.0, 2.0, n) * amp / dkk / dll / np.abs(dmm)
    return dict(dens=dens, lam=lam, phi=phi, rr=rr, drr=drr, kk=kk, ll=ll, mm=mm,
                dmm=dmm, dkk=dkk, dll=dll, area=area)


STATE_KEYS = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm"]


def pack(ic, uu, vv):
    return obj_state([ic[k] for k in STATE_KEYS] + [uu, vv])


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes")


def flat_state(prefix, st):
    names = STATE_KEYS + ["uu", "vv"]
    return {f"{prefix}_{n}": np.asarray(st[i], dtype=np.float64) for i, n in enumerate(names)}


# ---------------------------------------------------------------- G1
def gen_g1():
    rng = np.random.default_rng(20240101)
    for tag, phi0 in (("f0", 0.0), ("f45", np.deg2rad(45.0))):
        for sat in (False, True):
            grid, grids, uu, vv = configure(phi0=phi0, kappa=1.0, saturate_online=sat)
            # a non-trivial v column so that dv/dz matters
            vv = 0.5 * uu[::-1].copy()
            lprop.set_pressure_gradient(uu, vv)
            ic = random_rays(rng, 257, grids, phi0, amp_scale=1.0)
            st = pack(ic, uu, vv)
            out = lprop.rhs_default(120.0, st)
            d = {}
            d.update(flat_state("in", st))
            d.update(flat_state("out", out))
            d.update(dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
                     pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy(),
                     grid=grid, dt=120.0, phi0=phi0, kappa=1.0, bvf=0.01,
                     saturate_online=int(sat))
            # also the raw flux profile the RHS used
            d["pm_flux_inner"] = lprop.wave_projection(
                ic["dens"], ic["lam"], ic["phi"], ic["rr"] - .5 * ic["drr"], ic["rr"] + .5 * ic["drr"],
                ic["kk"], ic["ll"], ic["mm"] - .5 * ic["dmm"], ic["mm"] + .5 * ic["dmm"],
                ic["dkk"], ic["dll"], ic["dmm"], grids)
            save(f"g1_rhs_{tag}_sat{int(sat)}", **d)


# ---------------------------------------------------------------- G2
def gen_g2():
    rng = np.random.default_rng(20240202)
    grid, grids, uu, vv = configure(ngrid=11, grid_max=10e3)
    # edge-case table on a 10-point G: [lo, up]
    edges = np.array([[1200., 2700.], [1200., 1400.], [-300., 400.], [7200., 9800.],
                      [-900., -100.], [8200., 8800.], [9200., 12000.], [500., 1500.],
                      [0., 1000.], [2500., 2500.], [3000., 4000.], [-5000., 20000.],
                      [999.9999999, 1000.0000001], [8000., 8000.5], [7999.5, 8000.]])
    n = len(edges)
    one = np.ones(n)
    d = {}
    G10 = np.linspace(0, 9000, 10)
    for gname, G in (("G10", G10), ("grid", grid), ("grids", grids)):
        for var in (3, 4):                              # interface variants (:199-219): whole table only
            d[f"edge_{gname}_var{var}"] = lprop.wave_projection(
                one.copy(), 0 * one, 0 * one, edges[:, 0].copy(), edges[:, 1].copy(), 2e-4 * one, 1e-4 * one,
                -1e-3 * one, -1e-3 * one, one, one, one, G, var=var)
        for var in (0, 1, 2):
            out = lprop.wave_projection(one.copy(), 0 * one, 0 * one, edges[:, 0].copy(), edges[:, 1].copy(),
                                        2e-4 * one, 1e-4 * one, -1e-3 * one, -1e-3 * one,
                                        one, one, one, G, var=var)
            d[f"edge_{gname}_var{var}"] = out
            # per-ray rows (one ray at a time) for the weight table
            rows = []
            for i in range(n):
                s = slice(i, i + 1)
                rows.append(lprop.wave_projection(one[s].copy(), 0 * one[s], 0 * one[s], edges[s, 0].copy(),
                                                  edges[s, 1].copy(), 2e-4 * one[s], 1e-4 * one[s],
                                                  -1e-3 * one[s], -1e-3 * one[s], one[s], one[s], one[s],
                                                  G, var=var))
            d[f"edgerows_{gname}_var{var}"] = np.array(rows)
    d.update(edges=edges, G10=G10, grid11=grid, grids11=grids)

    grid, grids, uu, vv = configure(ngrid=101, grid_max=100e3)
    ic = random_rays(rng, 1000, grids, 0.3)
    for gname, G in (("grid", grid), ("grids", grids)):
        for var in (0, 1, 2, 3, 4):
            d[f"rand_{gname}_var{var}"] = lprop.wave_projection(
                ic["dens"], ic["lam"], ic["phi"], ic["rr"] - .5 * ic["drr"], ic["rr"] + .5 * ic["drr"],
                ic["kk"], ic["ll"], ic["mm"] - .5 * ic["dmm"], ic["mm"] + .5 * ic["dmm"],
                ic["dkk"], ic["dll"], ic["dmm"], G, var=var)
    for k in STATE_KEYS + ["dkk", "dll", "area"]:
        d["rand_" + k] = ic[k]
    d.update(grid101=grid, bvf=0.01)
    save("g2_projection", **d)


# ---------------------------------------------------------------- G3
def rhs_fixed_background(dt, var):
    out = lprop.rhs_default(dt, var)
    out[9] = np.zeros_like(out[9])
    out[10] = np.zeros_like(out[10])
    return out


def run_steps(st, dt, marks):
    res = {}
    for n in range(1, max(marks) + 1):
        st = lprop.RK3(dt, st)
        if n in marks:
            res[n] = [np.array(s, dtype=np.float64) for s in st]
    return res


def gen_g3():
    dt = 120.0
    # coupled, driver regime
    grid, grids, uu, vv = configure(ngrid=101)
    ic = driver_ic(60, grids, alpha=0.01)
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    for n, s in run_steps(st, dt, (1, 10, 100)).items():
        d.update(flat_state(f"s{n}", s))
    save("g3_rk3_coupled_driver", **d)

    # coupled with Coriolis (phi0 = 45 deg) and a v column: exercises f0*v, pg terms
    phi0 = np.deg2rad(45.0)
    grid, grids, uu, vv = configure(ngrid=101, phi0=phi0)
    vv = 0.25 * uu[::-1].copy()
    lprop.set_pressure_gradient(uu, vv)
    ic = driver_ic(60, grids, alpha=0.05, phi0=phi0)
    # spread the azimuth so that both flux components are non-zero
    az = np.linspace(0, 2 * np.pi, 60, endpoint=False)
    kh = 2 * np.pi / 50e3
    ic["kk"] = kh * np.sin(az)
    ic["ll"] = kh * np.cos(az)
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=phi0, kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    for n, s in run_steps(st, dt, (1, 10, 100)).items():
        d.update(flat_state(f"s{n}", s))
    save("g3_rk3_coupled_f45", **d)

    # fixed background = BASELINE config 1 (100 rays, ngrid 201, 1000 steps)
    grid, grids, uu, vv = configure(ngrid=201, rhs=rhs_fixed_background)
    ic = driver_ic(100, grids, alpha=0.01)
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    for n, s in run_steps(st, dt, (1, 10, 100, 1000)).items():
        d.update(flat_state(f"s{n}", s))
    save("g3_rk3_fixedbg_config1", **d)
    lprop.set_model_setup(rhs=lprop.rhs_default)


# ---------------------------------------------------------------- G4
def gen_g4():
    dt = 120.0
    rng = np.random.default_rng(20240404)
    # online saturation, strong amplitude
    grid, grids, uu, vv = configure(ngrid=101, kappa=1.0, saturate_online=True)
    ic = random_rays(rng, 777, grids, 0.0, amp_scale=0.6, zlo=500., zhi=60e3)
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=1,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    res = run_steps(st, dt, (1, 5, 20, 60))
    for n, s in res.items():
        d.update(flat_state(f"s{n}", s))
    d["n_changed_dens_s60"] = int(np.sum(res[60][0] != ic["dens"]))
    save("g4_saturation_online", **d)
    print("   online saturation: rays whose dens changed by step 60:", d["n_changed_dens_s60"])

    # online saturation on a FIXED background (no coupling -> not chaotic): long horizon
    grid, grids, uu, vv = configure(ngrid=101, kappa=1.0, saturate_online=True, rhs=rhs_fixed_background)
    ic = random_rays(rng, 777, grids, 0.0, amp_scale=0.6, zlo=500., zhi=60e3)
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=1,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    res = run_steps(st, dt, (1, 20, 60))
    for n, s in res.items():
        d.update(flat_state(f"s{n}", s))
    d["n_changed_dens_s60"] = int(np.sum(res[60][0] != ic["dens"]))
    save("g4_saturation_online_fixedbg", **d)
    print("   online saturation (fixed bg): rays whose dens changed by step 60:", d["n_changed_dens_s60"])
    lprop.set_model_setup(rhs=lprop.rhs_default)

    # direct saturation: the driver's own loop (raytracer.py:157-188)
    grid, grids, uu, vv = configure(ngrid=101, kappa=1.0, saturate_online=False)
    ic = driver_ic(60, grids, alpha=0.01)
    nt_max = 1440
    rows = (1, 10, 100, 709, 710, 711, 1000, 1440)
    cur = {k: ic[k].copy() for k in STATE_KEYS}
    cur_uu, cur_vv = uu.copy(), vv.copy()
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", pack(ic, uu, vv)))
    nsat = 0
    for nt in range(1, nt_max + 1):
        st_in = pack(cur, cur_uu, cur_vv)
        out = lprop.RK3(dt, st_in)
        new = {k: np.array(out[i], dtype=np.float64) for i, k in enumerate(STATE_KEYS)}
        dens_prop = new["dens"]
        dens_sat = lprop.saturation(
            dt, dens_prop, cur["rr"], (new["rr"] - cur["rr"]) / 1,
            cur["drr"], (new["drr"] - cur["drr"]) / dt,
            new["kk"], new["ll"], cur["mm"], (new["mm"] - cur["mm"]) / dt, direct=True)
        nsat += int(np.sum(dens_sat != dens_prop))
        new["dens"] = dens_sat
        cur = new
        cur_uu, cur_vv = np.array(out[9], dtype=np.float64), np.array(out[10], dtype=np.float64)
        if nt in rows:
            d.update(flat_state(f"s{nt}", [cur[k] for k in STATE_KEYS] + [cur_uu, cur_vv]))
            d[f"s{nt}_dens_prop"] = dens_prop
            # the driver's conservation diagnostic on this stored row (raytracer.py:198-240): wave action
            # on `grid` (var=2, :213) and vertical wave-action flux on `grids` (var=1, :227)
            lo, up = cur["rr"] - .5 * cur["drr"], cur["rr"] + .5 * cur["drr"]
            mlo, mup = cur["mm"] - .5 * cur["dmm"], cur["mm"] + .5 * cur["dmm"]
            d[f"s{nt}_wa"] = lprop.wave_projection(cur["dens"], cur["lam"], cur["phi"], lo, up, cur["kk"], cur["ll"],
                                                   mlo, mup, ic["dkk"], ic["dll"], cur["dmm"], grid, var=2)
            d[f"s{nt}_flux_diag"] = lprop.wave_projection(cur["dens"], cur["lam"], cur["phi"], lo, up, cur["kk"],
                                                          cur["ll"], mlo, mup, ic["dkk"], ic["dll"], cur["dmm"],
                                                          grids, var=1)
    d["n_saturation_events"] = nsat
    save("g4_saturation_direct_driver", **d)
    print("   driver loop saturation events:", nsat)


# ---------------------------------------------------------------- G5
def gen_g5():
    """The bench workload at toy size: synthetic Gaussian spectrum, coupled."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "python-msgwam_amd"))
    from msgwam_amd.spectrum import gaussian_spectrum  # host-side numpy only
    dt = 120.0
    grid, grids, uu, vv = configure(ngrid=101)
    sp = gaussian_spectrum(2000, grids, lprop.rhobar, alpha=0.01, nz=20, nd=4)
    ic = {k: sp[k] for k in STATE_KEYS}
    lprop.set_statics(dll=sp["dll"], dkk=sp["dkk"], rr_mm_area=sp["area"])
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=0.0, kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=sp["dkk"], dll=sp["dll"], area=sp["area"],
             pg=lprop.pressure_gradient.copy(), rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    for n, s in run_steps(st, dt, (1, 3)).items():
        d.update(flat_state(f"s{n}", s))
    save("g5_spectrum_coupled", **d)


# ---------------------------------------------------------------- G6
def gen_g6():
    """HPROP_GLOBAL = True (libprop's own default): lam, phi, kk, ll evolve too."""
    rng = np.random.default_rng(20240606)
    dt = 120.0
    for tag, sat in (("sat0", False), ("sat1", True)):
        grid, grids, uu, vv = configure(phi0=np.deg2rad(30.0), kappa=1.0, saturate_online=sat)
        vv = 0.5 * uu[::-1].copy()
        lprop.set_pressure_gradient(uu, vv)
        lprop.HPROP_GLOBAL = True
        ic = random_rays(rng, 257, grids, np.deg2rad(30.0), amp_scale=1.0)
        ic["phi"] = rng.uniform(np.deg2rad(-70.0), np.deg2rad(70.0), 257)      # rays at scattered latitudes
        ic["lam"] = rng.uniform(0.0, 2 * np.pi, 257)
        st = pack(ic, uu, vv)
        out = lprop.rhs_default(dt, st)
        d = {}
        d.update(flat_state("in", st))
        d.update(flat_state("out", out))
        d.update(dkk=ic["dkk"], dll=ic["dll"], area=ic["area"], pg=lprop.pressure_gradient.copy(),
                 rhobar=lprop.rhobar.copy(), grid=grid, dt=dt, phi0=np.deg2rad(30.0), kappa=1.0, bvf=0.01,
                 saturate_online=int(sat))
        save(f"g6_hprop_rhs_{tag}", **d)
    # RK3 with horizontal propagation, coupled, driver-like amplitudes
    grid, grids, uu, vv = configure(phi0=np.deg2rad(30.0))
    vv = 0.5 * uu[::-1].copy()
    lprop.set_pressure_gradient(uu, vv)
    lprop.HPROP_GLOBAL = True
    ic = driver_ic(60, grids, alpha=0.01, phi0=np.deg2rad(30.0))
    ic["phi"] = np.deg2rad(30.0) + np.linspace(-0.2, 0.2, 60)
    ic["kk"] = ic["kk"] * np.linspace(0.5, 1.5, 60)
    ic["ll"] = ic["kk"][::-1] * 0.7 + 1e-5
    st = pack(ic, uu, vv)
    d = dict(grid=grid, dt=dt, phi0=np.deg2rad(30.0), kappa=1.0, bvf=0.01, saturate_online=0,
             dkk=ic["dkk"], dll=ic["dll"], area=ic["area"], pg=lprop.pressure_gradient.copy(),
             rhobar=lprop.rhobar.copy())
    d.update(flat_state("in", st))
    for n, s_ in run_steps(st, dt, (1, 5, 20, 100)).items():
        d.update(flat_state(f"s{n}", s_))
    save("g6_hprop_rk3_coupled", **d)
    lprop.HPROP_GLOBAL = False


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6"]
    np.seterr(all="ignore")
    for w in which:
        globals()["gen_" + w]()
