"""
ORACLE -- test infrastructure, NOT the product.

CPU (numpy) restatement of the batched ray-propagation hot path of
python-msgwam (`lib/libprop.py` + the step wrapper of `raytracer.py`).  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module; the product path (`python-msgwam_amd/`) never does
and fails loudly when its HIP extension is missing.

Parity is PINNED: every function below is checked in `tests/test_oracle_golden.py`
against fixtures in `tests/golden/*.npz` that `oracle/gen_golden.py` produced by
importing the real reference (`/root/reference/lib/libprop.py`) in the build
container (numpy 2.2.6).  Citations are `file:line` relative to the reference
root.

Design: the oracle keeps NO module-global state (the reference does,
`lib/libprop.py:3-11`); everything lives in a `Setup` object.  Arithmetic is
written operation by operation in the reference's order so that results are
bit-comparable with numpy's evaluation of the reference expressions.
"""
from __future__ import annotations

import numpy as np

RAD_EARTH = 6378e3      # lib/libprop.py:3
ROT_EARTH = 7.2921e-5   # lib/libprop.py:4

# Williamson low-storage RK3 coefficients exactly as the reference writes them
# (python float expressions, lib/libprop.py:693-698)
RK_A = (0.0, 5 / 9, 153 / 128)          # q = dt*f - A*q
RK_B15_16 = 15 / 16
RK_B8_15 = 8 / 15


class Setup:
    """Everything the reference keeps in module globals (lib/libprop.py:3-11,
    :14-44): config scalars, the column grids and the per-ray statics."""

    def __init__(self, grid, bvf=0.01, phi0=0.0, kappa=1.0, saturate_online=False,
                 hh=8500.0, rhobar0=1.2, boussinesq=False, dkk=None, dll=None,
                 rr_mm_area=None, hprop=False):
        self.grid = np.asarray(grid, dtype=np.float64)
        self.grids = .5 * (self.grid[:-1] + self.grid[1:])       # raytracer.py:75
        self.bvf = bvf
        self.phi0 = phi0
        self.kappa = kappa
        self.saturate_online = saturate_online
        self.hh = hh
        self.rhobar0 = rhobar0
        self.boussinesq = boussinesq
        self.dkk = dkk
        self.dll = dll
        self.rr_mm_area = rr_mm_area
        self.hprop = hprop                                       # lib/libprop.py:5 HPROP_GLOBAL
        # lib/libprop.py:47-62 set_hydrostatics
        if boussinesq:
            self.rhobar = rhobar0 * np.ones(self.grids.shape)
        else:
            self.rhobar = rhobar0 * np.exp(-self.grids / hh)
        self.pressure_gradient = np.zeros((2, len(self.grids)))

    def set_pressure_gradient(self, uu, vv):
        """lib/libprop.py:65-82"""
        ff = 2 * ROT_EARTH * np.sin(self.phi0)
        pg = np.empty((2, len(self.grids)))
        pg[0] = self.rhobar * ff * vv
        pg[1] = - self.rhobar * ff * uu
        self.pressure_gradient = pg


def velocities_sine_homogeneous(rr, u0, rr0, sig_rr):
    """lib/libprop.py:306-325 (host-side setup of the column only)."""
    exponential = .5 * (np.tanh((rr - rr0) / sig_rr) + 1)
    return u0 * exponential * np.sin(rr / sig_rr * 2 * np.pi)


# --------------------------------------------------------------------------
# a-4  dispersion relation and vertical group velocity
# --------------------------------------------------------------------------
def omega(kk, ll, mm, phi, bvf):
    """lib/libprop.py:369-383"""
    ff = 2 * ROT_EARTH * np.sin(phi)
    return np.sqrt((bvf ** 2 * (kk ** 2 + ll ** 2) + ff ** 2 * mm ** 2)
                   / (kk ** 2 + ll ** 2 + mm ** 2))


def cg_rr(kk, ll, mm, phi, bvf):
    """lib/libprop.py:434-448 (lam, rr are unused there)."""
    vk_square = kk ** 2 + ll ** 2 + mm ** 2
    ff = 2 * ROT_EARTH * np.sin(phi)
    om = omega(kk, ll, mm, phi, bvf)
    return - mm * (om ** 2 - ff ** 2) / om / vk_square


# --------------------------------------------------------------------------
# EXTENSION (north_star "U(z)/N2(z) column"): buoyancy frequency as a COLUMN on `grids`.
# The reference has a scalar `bvf` only (lib/libprop.py:380, :398, :422, :583), so this is build-defined and its
# parity is UNPINNED, except in the limit N(z) = const, where every expression below reduces to the reference's
# and the results must be bit-identical (tests/test_oracle_golden.py).  Definition: wherever the reference reads
# model_config['bvf'], N is np.interp'ed from the column to the height that expression is about --
#   cg_rr(..., rr +- drr/2)  (:635-636)         N at rr +- drr/2   => ddrr_st = cgr_up - cgr_down != 0 (:641), and
#                                               drr, dmm evolve (ddmm_st = dmm / drr * ddrr_st, :645)
#   cg_rr in wave_projection (:139-144)         N at the ray centre .5 * (rr_low + rr_up)
#   omega(kk, ll, mm_center, phi0) (:597)       N at rr_center;   NN**2 of the cap (:601): N at rr_final
# -- and dm/dt gets the refraction term of the WKB ray equations that a height-dependent N requires,
#   dm/dt -= (d omega / d N) * dN/dz = N * (k**2 + l**2) / (omega * |k|**2) * dN/dz   at the ray centre,
# with dN/dz the derivative of that very interpolant (the slope of the np.interp segment the ray centre is in, 0
# outside [grids[0], grids[-1]] where np.interp clamps), so that the intrinsic frequency is conserved along a ray in a
# steady N(z) up to the time-stepping error (the KAT of the tests).  NOTE the reference's ddmm_st = dmm / drr * ddrr_st
# (:645) keeps dmm / drr constant, not the area drr * dmm; it is reproduced as written.
# --------------------------------------------------------------------------
def bvf_at(setup, z):
    """N at height z: the scalar itself (reference), or np.interp on `grids` (end-value clamping)."""
    b = setup.bvf
    return b if np.ndim(b) == 0 else np.interp(z, setup.grids, np.asarray(b, dtype=np.float64))


def bvf_gradient_at(setup, z):
    """dN/dz at height z for a column N: the slope of the np.interp segment grids[j] <= z < grids[j+1] (numpy's own
    slope expression), 0 where np.interp clamps (z < grids[0] or z >= grids[-1]); NaN stays NaN."""
    b = np.asarray(setup.bvf, dtype=np.float64)
    g = setup.grids
    z = np.asarray(z, dtype=np.float64)
    slope = (b[1:] - b[:-1]) / (g[1:] - g[:-1])
    j = np.clip(np.searchsorted(g, z, side="right") - 1, 0, len(g) - 2)
    inside = (z >= g[0]) & (z < g[-1])
    return np.where(np.isnan(z), np.nan, np.where(inside, slope[j], 0.0))


# --------------------------------------------------------------------------
# a-5  shear interpolation
# --------------------------------------------------------------------------
def shear_at_rays(setup, rr, uu, vv):
    """lib/libprop.py:349-356: first differences on the interior interfaces and
    np.interp (piecewise linear, end-value clamping) to the ray heights."""
    grid = setup.grid
    dz = np.diff(grid[:2])[0]
    du_dz = (uu[1:] - uu[:-1]) / dz
    dv_dz = (vv[1:] - vv[:-1]) / dz
    return (np.interp(rr, grid[1:-1], du_dz), np.interp(rr, grid[1:-1], dv_dz))


# --------------------------------------------------------------------------
# a-3  wave_projection  (var = 0, 1, 2)
# --------------------------------------------------------------------------
def _projection_indices(rr_low, rr_up, G):
    """lib/libprop.py:123-135.  Returns dz, nlow, nup (clipped), skip mask."""
    dz = np.diff(G[:2])[0]
    with np.errstate(invalid='ignore', over='ignore'):
        nlow = (rr_low / dz).astype(int)            # trunc toward zero
        nup = (rr_up / dz + 1.).astype(int)
    nzmax = len(G) - 2
    ood = (((nlow >= nzmax) & (nup >= nzmax)) | ((nlow <= 0) & (nup <= 0)))
    nlow = np.clip(nlow, 0, nzmax)                  # :133-134 (on raw indices)
    nup = np.clip(nup, 0, nzmax)
    return dz, nlow, nup, ood


def wave_projection(dens, rr_low, rr_up, kk, ll, mm_low, mm_up, phi,
                    dkk, dll, dmm, G, bvf, var=0, loop=False):
    """lib/libprop.py:92-219 for var in {0, 1, 2, 3, 4}.

    `loop=False` is a vectorised closed form that accumulates in exactly the
    reference's order (ray-major, then cell), so it is bit-identical to
    `loop=True`, which keeps the reference's interpreted double loop
    (:151-163) and is what `bench.py` times as the 1-core CPU baseline.
    """
    G = np.asarray(G, dtype=np.float64)
    dz, nlow, nup, ood = _projection_indices(rr_low, rr_up, G)
    phase_space_vol = abs(dkk * dll * dmm)                       # :137
    cgr = cg_rr(kk, ll, .5 * (mm_low + mm_up), phi, bvf)         # :139-144
    if var == 0:
        payload = (cgr * kk * dens, cgr * ll * dens)             # :148-149
    elif var == 1:
        payload = (cgr * dens,)                                  # :167
    elif var == 2:
        payload = (dens,)                                        # :184
    elif var in (3, 4):
        # :199-219: wave-action flux (3) / pseudo-momentum fluxes (4) at the INTERFACES nb = 1 .. len(G)-2: the sum
        # over the rays straddling nb (nlow < nb < nup on the clipped indices), np.sum per interface as there
        payload = (cgr * dens,) if var == 3 else (cgr * kk * dens, cgr * ll * dens)
        out = np.zeros((len(payload), len(G)))
        live = ~ood
        for nb in range(1, len(G) - 1):
            index = np.where((nlow < nb) & (nup > nb) & live)
            for p, v in enumerate(payload):
                out[p, nb] += np.sum(v[index] * phase_space_vol[index])
        return out[0] if var == 3 else out
    else:
        raise ValueError("var must be 0 .. 4")
    ncell = len(G) - 1
    out = np.zeros((len(payload), ncell))

    if loop:
        for nr in range(len(dens)):
            if ood[nr]:
                continue
            for ncell_ in range(nlow[nr], nup[nr]):
                zmin = np.max([G[ncell_], rr_low[nr]])
                zmax = np.min([G[ncell_ + 1], rr_up[nr]])
                dri_o_dr = np.abs(zmax - zmin) / dz
                for p, v in enumerate(payload):
                    out[p, ncell_] += dri_o_dr * phase_space_vol[nr] * v[nr]
    else:
        cnt = np.where(ood, 0, np.maximum(nup - nlow, 0))
        total = int(cnt.sum())
        if total:
            ray = np.repeat(np.arange(len(dens)), cnt)
            first = np.cumsum(cnt) - cnt
            cell = nlow[ray] + (np.arange(total) - first[ray])
            zmin = np.maximum(G[cell], rr_low[ray])
            zmax = np.minimum(G[cell + 1], rr_up[ray])
            w = np.abs(zmax - zmin) / dz
            for p, v in enumerate(payload):
                # bincount adds sequentially in array order == ray order
                out[p] = np.bincount(cell, weights=w * phase_space_vol[ray] * v[ray],
                                     minlength=ncell)
    return out if var == 0 else out[0]


# --------------------------------------------------------------------------
# a-7  saturation
# --------------------------------------------------------------------------
def saturation(setup, dt, dens, rr_center, rr_center_st, drr, drr_st, kk, ll,
               mm_center, mm_center_st, direct=False):
    """lib/libprop.py:561-615"""
    phi0, kappa = setup.phi0, setup.kappa
    ff = 2 * ROT_EARTH * np.sin(phi0)
    rr_final = rr_center + rr_center_st * dt
    drr_final = drr + drr_st * dt
    mm_final = mm_center + mm_center_st * dt
    dmm_final = setup.rr_mm_area / drr_final
    rhobar_final = np.interp(rr_final, setup.grids, setup.rhobar)
    NN = bvf_at(setup, rr_final)                         # scalar bvf: the reference's NN (:583)
    omh = omega(kk, ll, mm_center, phi0, bvf_at(setup, rr_center))
    phase_volume = setup.dkk * setup.dll * dmm_final
    with np.errstate(divide='ignore', invalid='ignore'):
        max_dens_final = (kappa ** 2 * .5 * rhobar_final * omh * NN ** 2
                          / mm_final ** 2 / (omh ** 2 - ff ** 2))
    hit = max_dens_final < dens * phase_volume
    if direct:
        dens_new = dens.copy()
        dens_new[hit] = max_dens_final[hit]
        return dens_new
    dens_st = np.zeros(dens.shape)
    dens_st[hit] = (max_dens_final[hit] - dens[hit]) / dt
    return dens_st


# --------------------------------------------------------------------------
# a-2 / a-6  right-hand side  (HPROP_GLOBAL = False branch, scalar bvf)
# --------------------------------------------------------------------------
def rhs(setup, dt, state, loop=False, fixed_background=False, return_flux=False, flux_reduce=None):
    """lib/libprop.py:618-676; HPROP_GLOBAL=False (raytracer.py:38) unless setup.hprop.

    state = [dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv].
    `fixed_background=True` is the rhs-hook variant of BASELINE config 1/2:
    identical tendencies with slots 9, 10 zeroed.
    `flux_reduce` (multi-rank tests only): callable applied to this rank's
    (2, ngrid-2) projection, e.g. a gloo all-reduce over ray shards.
    """
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = state
    ncol = np.ndim(setup.bvf) != 0                       # EXTENSION: N as a column on grids (see bvf_at)
    bvf = bvf_at(setup, rr)                              # scalar bvf: the scalar itself
    cgr_up = cg_rr(kk, ll, mm, phi, bvf_at(setup, rr + .5 * drr))     # :635 (rr unused there: scalar bvf)
    cgr_down = cg_rr(kk, ll, mm, phi, bvf_at(setup, rr - .5 * drr))   # :636
    zeros = np.zeros(np.shape(kk))
    du_dz_ray, dv_dz_ray = shear_at_rays(setup, rr, uu, vv)
    gradient = (kk * du_dz_ray + ll * dv_dz_ray)         # :517
    if ncol:                                             # refraction by dN/dz (extension, see above)
        gradient = gradient + (bvf * (kk ** 2 + ll ** 2) / omega(kk, ll, mm, phi, bvf)
                               / (kk ** 2 + ll ** 2 + mm ** 2) * bvf_gradient_at(setup, rr))
    if setup.hprop:
        # HPROP_GLOBAL = True (lib/libprop.py:5): horizontal propagation on the sphere
        om = omega(kk, ll, mm, phi, bvf)
        vk_square = kk ** 2 + ll ** 2 + mm ** 2
        uu_ray = np.interp(rr, setup.grids, uu)          # :399, :357
        vv_ray = np.interp(rr, setup.grids, vv)          # :422, :358
        cg_lam = kk / om / vk_square * (bvf ** 2 - om ** 2) + uu_ray     # :404
        cg_ph = ll / om / vk_square * (bvf ** 2 - om ** 2) + vv_ray      # :428
        cgr = cg_rr(kk, ll, mm, phi, bvf)
        dlam_st = cg_lam / (RAD_EARTH + rr) / np.cos(phi)                # :638
        dphi_st = cg_ph / (RAD_EARTH + rr)                               # :639
        grad_k = (kk * zeros + ll * zeros) / (RAD_EARTH + rr) / np.cos(phi)   # :463-464 (no horizontal wind gradients)
        dkk_st = kk / (RAD_EARTH + rr) * (np.tan(phi) * cg_ph - cgr) - grad_k   # :467-468
        grad_l = (kk * zeros + ll * zeros) / (RAD_EARTH + rr)            # :487
        df2_dphi = 8 * ROT_EARTH ** 2 * np.sin(phi) * np.cos(phi) * 1    # :489
        dll_st = - (ll * cgr + kk * np.tan(phi) * cg_lam
                    + mm ** 2 / 2 / om / (kk ** 2 + ll ** 2 + mm ** 2) * df2_dphi) / (RAD_EARTH + rr) \
            - grad_l                                                     # :492-496
        dmm_st = (kk * cg_lam + ll * cg_ph) / (RAD_EARTH + rr) - gradient   # :519-520
    else:
        # :638-639  cg_lambda/cg_phi return zeros when HPROP is off
        dlam_st = zeros / (RAD_EARTH + rr) / np.cos(phi)
        dphi_st = zeros / (RAD_EARTH + rr)
        dkk_st = np.zeros(np.shape(kk))                  # :470-471
        dll_st = np.zeros(np.shape(kk))                  # :498-499
        dmm_st = (kk * zeros + ll * zeros) / (RAD_EARTH + rr) - gradient   # :519-520
    drr_st = .5 * (cgr_down + cgr_up)                    # :640
    ddrr_st = cgr_up - cgr_down                          # :641
    ddmm_st = dmm / drr * ddrr_st                        # :645
    dens_st = setup.saturate_online * saturation(        # :647-651
        setup, dt, dens, rr, drr_st, drr, ddrr_st, kk, ll, mm, dmm_st)

    grid, grids = setup.grid, setup.grids
    pm_flux = np.zeros((2, len(grid)))                   # :653
    proj = wave_projection(                              # :654-658
        dens, rr - .5 * drr, rr + .5 * drr, kk, ll, mm - .5 * dmm, mm + .5 * dmm,
        phi, setup.dkk, setup.dll, dmm, grids, bvf, var=0, loop=loop)
    if flux_reduce is not None:
        proj = flux_reduce(proj)
    pm_flux[:, 1:-1] = proj
    pm_flux[:, 0] = pm_flux[:, 1]
    pm_flux[:, -1] = pm_flux[:, -2]
    dz = np.diff(grid[:2])[0]
    pm_flux_gradient = (pm_flux[:, 1:] - pm_flux[:, :-1]) / dz     # :663

    ff0 = 2 * ROT_EARTH * np.sin(setup.phi0)
    pg = setup.pressure_gradient
    du_st = ff0 * vv - setup.rhobar ** -1 * (pg[0] + pm_flux_gradient[0])   # :537
    dv_st = -ff0 * uu - setup.rhobar ** -1 * (pg[1] + pm_flux_gradient[1])  # :556
    if fixed_background:
        du_st = np.zeros_like(du_st)
        dv_st = np.zeros_like(dv_st)
    out = [dens_st, dlam_st, dphi_st, drr_st, ddrr_st, dkk_st, dll_st,
           dmm_st, ddmm_st, du_st, dv_st]
    if return_flux:
        return out, pm_flux
    return out


def rk3(setup, dt, state, loop=False, fixed_background=False, flux_reduce=None):
    """lib/libprop.py:680-700, slot by slot (so nray == ngrid-1 cannot collapse
    the object array as it does in the reference)."""
    var = [np.asarray(s, dtype=np.float64) for s in state]
    f = lambda v: rhs(setup, dt, v, loop=loop, fixed_background=fixed_background, flux_reduce=flux_reduce)
    qq = [dt * r for r in f(var)]
    var = [v + q / 3 for v, q in zip(var, qq)]
    qq = [dt * r - 5 / 9 * q for r, q in zip(f(var), qq)]
    var = [v + 15 / 16 * q for v, q in zip(var, qq)]
    qq = [dt * r - 153 / 128 * q for r, q in zip(f(var), qq)]
    var = [v + 8 / 15 * q for v, q in zip(var, qq)]
    return var


# --------------------------------------------------------------------------
# a-8  the driver's step wrapper
# --------------------------------------------------------------------------
def driver_step(setup, dt, state, loop=False, fixed_background=False,
                ref_quirks=True):
    """raytracer.py:157-188: RK3, then (if not saturate_online) the direct
    saturation with the driver's argument quirks: rr tendency divided by 1
    instead of dt (:184), new kk/ll, old mm (:186).
    Returns (state_out with saturated dens, dens_prop)."""
    old = [np.asarray(s, dtype=np.float64) for s in state]
    new = rk3(setup, dt, old, loop=loop, fixed_background=fixed_background)
    dens_prop = new[0]
    if not setup.saturate_online:
        rr_div = 1 if ref_quirks else dt
        new = list(new)
        new[0] = saturation(
            setup, dt, dens_prop, old[3], (new[3] - old[3]) / rr_div,
            old[4], (new[4] - old[4]) / dt, new[5], new[6], old[7],
            (new[7] - old[7]) / dt, direct=True)
    return new, dens_prop


# --------------------------------------------------------------------------
# EXTENSION (not in the reference; SURVEY 8f rank 2 / BASELINE config 5): removal +
# continuous source relaunch.  Parity UNPINNED: this function is the definition,
# the GPU path (MSGW_RELAUNCH) must match it.
# --------------------------------------------------------------------------
def relaunch(setup, state, source, frac=1e-6):
    """After a complete RK3 step (and after the direct saturation, when used) a ray
    slot is recycled to its SOURCE values of (dens, rr, mm) -- the values it was
    launched with -- when
      * its volume has left the column:  rr - drr/2 > grid[-1]  or  rr + drr/2 < grid[0]
        (such rays deposit nothing any more, lib/libprop.py:129-135), or
      * it has broken:  dens < frac * dens_source  (the reference's saturation
        collapses dens by ~12 orders of magnitude, lib/libprop.py:604-610).
    Comparisons with NaN are false (a NaN ray is left alone).  Everything else
    (drr, kk, ll, dmm, lam, phi, the column) is untouched.
    `source` = (dens, rr, mm) arrays.  Returns (new state list, mask of recycled rays)."""
    st = [np.asarray(s, dtype=np.float64) for s in state]
    dens, rr, drr, mm = st[0], st[3], st[4], st[7]
    sd, sr, sm = (np.asarray(a, dtype=np.float64) for a in source)
    z_bot, z_top = setup.grid[0], setup.grid[-1]
    with np.errstate(invalid="ignore"):
        out = (rr - .5 * drr > z_top) | (rr + .5 * drr < z_bot) | (dens < frac * sd)
    st[0] = np.where(out, sd, dens)
    st[3] = np.where(out, sr, rr)
    st[7] = np.where(out, sm, mm)
    return st, out
