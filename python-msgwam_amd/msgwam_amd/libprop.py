"""
Drop-in mirror of the module surface of python-msgwam's `lib/libprop.py` for the
ray-propagation hot path, backed by the MI355X HIP library (through
`_capi`, a ctypes binding of include/msgwam_hip.h).

    import msgwam_amd.libprop as lprop      # instead of `import lib.libprop as lprop`

Same names, argument orders and return shapes as the reference for everything
`raytracer.py` touches: the module attributes `HPROP_GLOBAL`,
`grid`, `grids`, `rhobar`, `pressure_gradient`, `model_config`, `statics`; the
setters; `omega`; and the hot-path entry points `RK3`, `rhs_default`,
`saturation`, `wave_projection`, which run on the GPU.  There is NO CPU
fallback for those four: without the HIP library or a GPU they raise.

Scope and deliberate differences (also in DESIGN.md):
  * Only a scalar `bvf` is implemented.  Both `HPROP_GLOBAL` branches run on the GPU:
    False (the driver's, raytracer.py:38) through the tuned kernels, True (lam, phi,
    kk, ll evolve as well) through a plain kernel of its own.
  * `model_config['rhs']` must be one of this module's built-ins (`rhs_default`,
    `rhs_fixed_background`).  An arbitrary Python callable is opaque host code
    that cannot run on the device; it raises TypeError instead of silently
    running a CPU path.
  * `RK3` builds its 11-slot object array slot by slot, so `nray == ngrid-1`
    works (the reference crashes there, lib/libprop.py:668-674).
Column/initial-condition helpers (`set_hydrostatics`, `set_pressure_gradient`,
`velocities_*`, `omega`) are one-off O(ngrid)/O(nray) host-side setup in numpy,
as in the reference.
"""
import numpy as np

from . import _capi

RAD_EARTH = 6378e3          # lib/libprop.py:3
ROT_EARTH = 7.2921e-5       # lib/libprop.py:4
HPROP_GLOBAL = True         # lib/libprop.py:5 (the driver switches it off, raytracer.py:38)
pressure_gradient = 0       # lib/libprop.py:6
grid = None                 # lib/libprop.py:7
grids = None                # lib/libprop.py:8
rhobar = 1                  # lib/libprop.py:9
model_config = {}           # lib/libprop.py:10
statics = {}                # lib/libprop.py:11


# ----------------------------------------------------------------------------
# configuration surface (lib/libprop.py:14-89)
# ----------------------------------------------------------------------------
def set_statics(**kwargs):
    """lib/libprop.py:14-27"""
    statics.update(kwargs)


def set_model_setup(**kwargs):
    """lib/libprop.py:30-44"""
    model_config.update(kwargs)


def get_model_setup():
    """lib/libprop.py:85-89"""
    return model_config


def set_hydrostatics():
    """lib/libprop.py:47-62: hydrostatic density on the staggered grid."""
    global rhobar
    if model_config['boussinesq']:
        rhobar = model_config['rhobar0'] * np.ones(grids.shape)
    else:
        rhobar = model_config['rhobar0'] * np.exp(-grids / model_config['hh'])


def set_pressure_gradient(uu, vv):
    """lib/libprop.py:65-82: geostrophically balanced pressure gradient."""
    global pressure_gradient
    ff = 2 * ROT_EARTH * np.sin(model_config['phi0'])
    pg = np.empty((2, len(grids)))
    pg[0] = rhobar * ff * vv
    pg[1] = - rhobar * ff * uu
    pressure_gradient = pg


def velocities_tanh_homogeneous(rr):
    """lib/libprop.py:253-273"""
    shape = (np.tanh((rr - model_config['rr0']) / model_config['sig_rr']) + 1) * 0.5
    return model_config['u0'] * shape


def velocities_gauss_homogeneous(rr):
    """lib/libprop.py:276-303 (its out-of-bounds mask can never be true; kept as is)."""
    u0, rr0, sig = model_config['u0'], model_config['rr0'], model_config['sig_rr']
    uu = u0 * np.exp(-(rr - rr0) ** 2 / 2 / sig ** 2)
    uu[np.where((rr <= rr0 - 3 * sig) & (rr >= rr0 + 3 * sig))] = 0.
    return uu


def velocities_sine_homogeneous(rr):
    """lib/libprop.py:306-325 (the driver's wind profile, raytracer.py:93)."""
    envelope = .5 * (np.tanh((rr - model_config['rr0']) / model_config['sig_rr']) + 1)
    return model_config['u0'] * envelope * np.sin(rr / model_config['sig_rr'] * 2 * np.pi)


def omega(kk, ll, mm, phi):
    """lib/libprop.py:369-383: intrinsic frequency (host-side; the driver uses it
    for the initial wave-action density, raytracer.py:114)."""
    bvf = model_config['bvf']
    ff = 2 * ROT_EARTH * np.sin(phi)
    return np.sqrt((bvf ** 2 * (kk ** 2 + ll ** 2) + ff ** 2 * mm ** 2) / (kk ** 2 + ll ** 2 + mm ** 2))



# ----------------------------------------------------------------------------
# The building blocks of the reference's rhs_default as host-side numpy helpers (same names, argument
# orders and module globals as lib/libprop.py).  They are here for scripts that call them directly, e.g.
# to plot group velocities; the GPU path has its own implementation and never calls them.
# ----------------------------------------------------------------------------
def _horizontal_norm(rr):
    return RAD_EARTH + rr


def gradients(lam_ray, phi_ray, rr_ray, uu, vv):
    """lib/libprop.py:328-366: winds and wind gradients at the rays, shape (4, 3) + rays.shape:
    [0] = (u, v, w), [1] = grad u, [2] = grad v, [3] = grad w in (lam, phi, r).  Only the vertical shear
    is non-zero in the 1-D column."""
    dz = np.diff(grid[:2])[0]
    out = np.zeros((4, 3) + np.shape(lam_ray))
    out[0, 0] = np.interp(rr_ray, grids, uu)
    out[0, 1] = np.interp(rr_ray, grids, vv)
    out[1, 2] = np.interp(rr_ray, grid[1:-1], (uu[1:] - uu[:-1]) / dz)
    out[2, 2] = np.interp(rr_ray, grid[1:-1], (vv[1:] - vv[:-1]) / dz)
    return out


def cg_rr(kk, ll, mm, lam, phi, rr):
    """lib/libprop.py:434-448: vertical group velocity."""
    ff = 2 * ROT_EARTH * np.sin(phi)
    om = omega(kk, ll, mm, phi)
    return - mm * (om ** 2 - ff ** 2) / om / (kk ** 2 + ll ** 2 + mm ** 2)


def _cg_horizontal(kh, kk, ll, mm, phi, wind_ray):
    if not HPROP_GLOBAL:                                   # :406, :430
        return np.zeros(np.shape(kk))
    bvf = model_config['bvf']
    om = omega(kk, ll, mm, phi)
    return kh / om / (kk ** 2 + ll ** 2 + mm ** 2) * (bvf ** 2 - om ** 2) + wind_ray


def cg_lambda(kk, ll, mm, lam, phi, rr, uu, vv):
    """lib/libprop.py:385-407: zonal group velocity (zero with HPROP_GLOBAL off)."""
    return _cg_horizontal(kk, kk, ll, mm, phi, np.interp(rr, grids, uu))


def cg_phi(kk, ll, mm, lam, phi, rr, uu, vv):
    """lib/libprop.py:409-431: meridional group velocity (zero with HPROP_GLOBAL off)."""
    return _cg_horizontal(ll, kk, ll, mm, phi, np.interp(rr, grids, vv))


def dk_dt(kk, ll, mm, lam, phi, rr, uu, vv):
    """lib/libprop.py:451-471: tendency of the zonal wavenumber (zero with HPROP_GLOBAL off)."""
    if not HPROP_GLOBAL:
        return np.zeros(np.shape(kk))
    vel = gradients(lam, phi, rr, uu, vv)
    gradient = (kk * vel[1, 0] + ll * vel[2, 0]) / _horizontal_norm(rr) / np.cos(phi)
    return kk / _horizontal_norm(rr) * (np.tan(phi) * cg_phi(kk, ll, mm, lam, phi, rr, uu, vv)
                                        - cg_rr(kk, ll, mm, lam, phi, rr)) - gradient


def dl_dt(kk, ll, mm, lam, phi, rr, uu, vv):
    """lib/libprop.py:474-499: tendency of the meridional wavenumber (zero with HPROP_GLOBAL off)."""
    if not HPROP_GLOBAL:
        return np.zeros(np.shape(kk))
    vel = gradients(lam, phi, rr, uu, vv)
    gradient = (kk * vel[1, 1] + ll * vel[2, 1]) / _horizontal_norm(rr)
    df2_dphi = 8 * ROT_EARTH ** 2 * np.sin(phi) * np.cos(phi) * 1
    return - (ll * cg_rr(kk, ll, mm, lam, phi, rr)
              + kk * np.tan(phi) * cg_lambda(kk, ll, mm, lam, phi, rr, uu, vv)
              + mm ** 2 / 2 / omega(kk, ll, mm, phi) / (kk ** 2 + ll ** 2 + mm ** 2) * df2_dphi) \
        / _horizontal_norm(rr) - gradient


def dm_dt(kk, ll, mm, lam, phi, rr, uu, vv):
    """lib/libprop.py:502-520: tendency of the vertical wavenumber."""
    vel = gradients(lam, phi, rr, uu, vv)
    gradient = kk * vel[1, 2] + ll * vel[2, 2]
    return (kk * cg_lambda(kk, ll, mm, lam, phi, rr, uu, vv)
            + ll * cg_phi(kk, ll, mm, lam, phi, rr, uu, vv)) / _horizontal_norm(rr) - gradient


def du_dt(vv, pm_flux_gradient):
    """lib/libprop.py:523-539: zonal mean-flow tendency."""
    ff = 2 * ROT_EARTH * np.sin(model_config['phi0'])
    return ff * vv - rhobar ** -1 * (pressure_gradient[0] + pm_flux_gradient)


def dv_dt(uu, pm_flux_gradient):
    """lib/libprop.py:542-558: meridional mean-flow tendency."""
    ff = 2 * ROT_EARTH * np.sin(model_config['phi0'])
    return -ff * uu - rhobar ** -1 * (pressure_gradient[1] + pm_flux_gradient)


def velocities_tanh(lam, phi, rr):
    """lib/libprop.py:222-246: jet, Gaussian in latitude and tanh in height; (4, 3) + lam.shape with the
    wind written into the whole first row, as there."""
    c = model_config
    shape = np.exp(-(phi - c['phi0']) ** 2 / 2 / c['sig_phi'] ** 2) * (np.tanh((rr - c['rr0']) / c['sig_rr']) + 1) * 0.5
    out = np.zeros((4, 3) + np.shape(lam))
    out[0] = c['u0'] * shape
    return out

# ----------------------------------------------------------------------------
# device backend
# ----------------------------------------------------------------------------
class _Backend:
    """One lazily created GPU context + what is resident in it."""

    def __init__(self):
        self.prop = None
        self.device = 0
        self.cfg = None         # (bvf, phi0, kappa, saturate_online)
        self.col = None         # copies of grid, grids, rhobar, pressure_gradient
        self.col_uv = None      # the (uu, vv) OBJECTS that are resident
        self.rays = None        # dict: input objects resident + statics objects
        self.out = None         # the arrays returned by the last RK3

    def context(self, ngrid, nray):
        p = self.prop
        if p is None or p.ngrid != ngrid or nray > p.cap:
            if p is not None:
                p.close()
            cap = max(int(nray), 1024)
            if p is not None and p.ngrid == ngrid:
                cap = max(cap, 2 * p.cap)
            self.prop = _capi.Propagator(ngrid, cap, device=self.device)
            self.cfg = self.col = self.col_uv = self.rays = self.out = None
        return self.prop

    def reset(self):
        if self.prop is not None:
            self.prop.close()
        self.__init__()


_backend = _Backend()


def set_device(device):
    """Select the HIP device of this process (one process per GPU)."""
    _backend.reset()
    _backend.device = int(device)


def release_device():
    """Free the GPU context (the next hot-path call recreates it)."""
    dev = _backend.device
    _backend.reset()
    _backend.device = dev


def _check_scope():
    if grid is None or grids is None:
        raise RuntimeError("lprop.grid / lprop.grids are not set (raytracer.py:76-77)")
    if np.ndim(model_config['bvf']) != 0:
        raise NotImplementedError("only a scalar bvf is supported (as in the reference)")


def _same(a, b):
    return a is b or (a is not None and b is not None and np.array_equal(a, b))


def _sync_config_and_column(p, uu, vv, force_uv):
    cfg = (float(model_config['bvf']), float(model_config['phi0']), float(model_config['kappa']),
           bool(model_config['saturate_online']), bool(HPROP_GLOBAL))
    if _backend.cfg != cfg:
        p.set_config(cfg[0], cfg[1], cfg[2], cfg[3], hprop=cfg[4])
        _backend.cfg = cfg
        _backend.rays = None                                     # lam, phi must be (re)uploaded with the rays
    col = _backend.col
    new = (np.asarray(grid, dtype=np.float64), np.asarray(grids, dtype=np.float64),
           np.asarray(rhobar, dtype=np.float64), np.asarray(pressure_gradient, dtype=np.float64))
    if np.shape(new[3]) != (2, len(new[1])):
        raise RuntimeError("lprop.pressure_gradient is not set: call set_pressure_gradient(uu, vv) "
                           "(raytracer.py:99)")
    stale = col is None or any(not _same(a, b) for a, b in zip(col, new))
    uv_resident = (not force_uv and not stale and _backend.col_uv is not None
                   and _backend.col_uv[0] is uu and _backend.col_uv[1] is vv)
    if not uv_resident:
        p.set_column(new[0], new[1], new[2], new[3], uu, vv)
        _backend.col = tuple(a.copy() for a in new)
        _backend.col_uv = (uu, vv)


_RAY_SLOTS = (0, 3, 4, 5, 6, 7, 8, 2)      # dens rr drr kk ll mm dmm phi


def _sync_rays(p, var):
    st = (statics['dkk'], statics['dll'], statics['rr_mm_area'])     # KeyError as in :585-587
    res = _backend.rays
    slots = _RAY_SLOTS + ((1,) if HPROP_GLOBAL else ())          # lam matters only with horizontal propagation
    resident = (res is not None and all(res['in'][i] is var[i] for i in slots)
                and all(a is b for a, b in zip(res['st'], st)))
    if not resident:
        dens, lam, phi, rr, drr, kk, ll, mm, dmm = [np.asarray(var[i], dtype=np.float64) for i in range(9)]
        p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, st[0], st[1], st[2])
        if HPROP_GLOBAL:
            p.upload_hprop(lam, phi)
    return not resident


def _prepare(var):
    _check_scope()
    if len(var) != 11:
        raise ValueError("the state vector must have 11 slots (lib/libprop.py:629)")
    nray = len(var[0])
    p = _backend.context(len(grid), nray)
    _sync_config_and_column(p, var[9], var[10], force_uv=False)
    _sync_rays(p, var)
    return p


def _pack(slots):
    out = np.empty(len(slots), dtype=object)
    for i, a in enumerate(slots):
        out[i] = a
    return out


def _flags_for(rhs):
    if rhs is rhs_default:
        return 0
    if rhs is rhs_fixed_background:
        return _capi.FIXED_BACKGROUND
    raise TypeError(
        "model_config['rhs'] must be msgwam_amd.libprop.rhs_default or .rhs_fixed_background; an "
        "arbitrary Python callable cannot run on the GPU and there is no CPU fallback")


# ----------------------------------------------------------------------------
# hot path (GPU)
# ----------------------------------------------------------------------------
def rhs_default(dt, var_in):
    """lib/libprop.py:618-676 on the GPU: the 11 tendencies of
    [dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv].  With HPROP off the
    tendencies of lam, phi, drr, kk, ll, dmm are identically zero; with HPROP on
    those of drr and dmm (ddrr_st = cgr_up - cgr_down = 0, :641, :645)."""
    return _rhs(dt, var_in, 0)


def rhs_fixed_background(dt, var_in):
    """Built-in rhs hook for a frozen mean flow (BASELINE configs 1, 2): the
    reference expresses it as a user hook that zeroes slots 9, 10 of
    rhs_default's result (lib/libprop.py:691)."""
    return _rhs(dt, var_in, _capi.FIXED_BACKGROUND)


def _rhs(dt, var_in, flags):
    p = _prepare(var_in)
    t = p.rhs(dt, flags)
    _backend.rays = dict(**{'in': list(var_in)}, st=(statics['dkk'], statics['dll'], statics['rr_mm_area']))
    _backend.col_uv = (var_in[9], var_in[10])
    z = lambda: np.zeros(np.shape(var_in[5]))
    if HPROP_GLOBAL:                                             # lam, phi, kk, ll have tendencies too (:638-643)
        lam, phi, kk, ll = p.download_hprop(tendencies=True)
        return _pack([t['dens'], lam, phi, t['rr'], z(), kk, ll, t['mm'], z(), t['uu'], t['vv']])
    return _pack([t['dens'], z(), z(), t['rr'], z(), z(), z(), t['mm'], z(), t['uu'], t['vv']])


def RK3(dt, var):
    """lib/libprop.py:680-700 on the GPU: one low-storage RK3 step of the full
    state (rays AND mean flow: uu, vv advance at every stage).  The state stays
    resident between calls when the caller passes back the arrays it was given."""
    flags = _flags_for(model_config['rhs'])
    p = _prepare(var)
    p.step(dt, 1, flags | _capi.NO_GRAPH)
    dens, rr, mm = p.download_rays()
    uu, vv = p.download_column()
    if HPROP_GLOBAL:
        lam, phi, kk, ll = p.download_hprop()
    else:
        lam, phi, kk, ll = (np.array(var[i], dtype=np.float64) for i in (1, 2, 5, 6))
    out = [dens, lam, phi, rr, np.array(var[4], dtype=np.float64), kk, ll, mm,
           np.array(var[8], dtype=np.float64), uu, vv]
    # what is now on the device corresponds to `out`; frozen slots are equal by value
    _backend.rays = dict(**{'in': list(out)}, st=(statics['dkk'], statics['dll'], statics['rr_mm_area']))
    _backend.col_uv = (uu, vv)
    return _pack(out)


def saturation(dt, dens, rr_center, rr_center_st, drr, drr_st, kk, ll, mm_center, mm_center_st,
               direct=False):
    """lib/libprop.py:561-615 on the GPU (the driver's post-step call,
    raytracer.py:182-188, uses direct=True)."""
    _check_scope()
    n = len(dens)
    p = _backend.context(len(grid), n)
    # the column may not be resident yet when the caller never ran RK3
    uv = _backend.col_uv or (np.zeros(len(grids)), np.zeros(len(grids)))
    _sync_config_and_column(p, uv[0], uv[1], force_uv=False)
    return p.saturation(dt, direct, dens, rr_center, rr_center_st, drr, drr_st, kk, ll, mm_center,
                        mm_center_st, statics['dkk'], statics['dll'], statics['rr_mm_area'])


def wave_projection(dens, lam, phi, rr_low, rr_up, kk, ll, mm_low, mm_up, dkk, dll, dmm, grid,
                    var=0):
    """lib/libprop.py:92-221 on the GPU: var 0, 1, 2 at the cell centres, var 3, 4 at the interfaces."""
    if var not in (0, 1, 2, 3, 4):
        raise ValueError("wave_projection: var must be 0 .. 4 (lib/libprop.py:97-102)")
    n = len(dens)
    ng = len(globals()['grid']) if globals()['grid'] is not None else len(grid)
    p = _backend.context(ng, n)
    return p.project_arrays(var, model_config['bvf'], dens, phi, rr_low, rr_up, kk, ll, mm_low, mm_up,
                            dkk, dll, dmm, grid)


# ----------------------------------------------------------------------------
# defaults, as at import of the reference (lib/libprop.py:703-726)
# ----------------------------------------------------------------------------
set_model_setup(
    u0=80, phi0=np.deg2rad(-60), sig_phi=np.deg2rad(3), rr0=30000, rr1=40000, sig_rr=10000, drr=1,
    bvf=0.01, rhs=rhs_default, geostrophy=True, boussinesq=False, hh=8500, rhobar0=1.2, kappa=0.95,
    saturate_online=True)
set_statics(int_dll=1, int_dkk=1, rr_mm_area=0)
