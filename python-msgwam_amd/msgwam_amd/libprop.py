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
  * `model_config['bvf']` is the reference's scalar or (EXTENSION) an array on `grids`.  Both
    `HPROP_GLOBAL` branches run on the GPU: False (the driver's, raytracer.py:38) through the tuned kernels, True
    (lam, phi, kk, ll evolve as well) and / or a bvf array (drr, dmm evolve as well) through the general stage kernel.
  * The reference's other building blocks (`gradients`, `cg_*`, `dk_dt` ... `dv_dt`, `velocities_tanh`) are NOT
    mirrored: they are internals of its `rhs_default` (no caller in `raytracer.py`), and a numpy copy of them here
    would be a CPU path beside the GPU one.
  * `model_config['rhs']` is honoured as in the reference (lib/libprop.py:691).  With the built-ins `rhs_default` /
    `rhs_fixed_background` a whole `RK3` step is one GPU call; any other callable (e.g. a hook around
    `lprop.rhs_default`) is driven by the reference's six RK lines on the host, its `rhs_default` calls on the GPU.
  * The evolving slots of the state `RK3` returns are `DeviceArray` objects: float64 arrays that live in HBM and are
    copied to the host on first access, so a loop that feeds the state back never crosses PCIe
    (`set_lazy_download(False)` returns plain ndarrays).  The unchanged slots are the caller's own objects.
    A slot counts as resident only if it is the same object as last time AND its content fingerprint is unchanged
    (`set_residency`), so an in-place edit between two calls is uploaded, as the reference would see it.
  * `RK3` builds its 11-slot object array slot by slot, so `nray == ngrid-1` works (the reference crashes there,
    lib/libprop.py:668-674).
Column/initial-condition helpers (`set_hydrostatics`, `set_pressure_gradient`,
`velocities_*`, `omega`) are one-off O(ngrid)/O(nray) host-side setup in numpy,
as in the reference.
"""
import weakref

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
# device backend
# ----------------------------------------------------------------------------
def _digest_stdlib(a):
    """Position-sensitive content digest without third-party modules (a swap of two rays, an in-place sort or any
    other reordering changes it): blake2b over the raw bytes."""
    import hashlib
    a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, hashlib.blake2b(memoryview(a).cast("B"), digest_size=16).digest())


try:                                                     # content digests of host arrays (residency checks)
    import xxhash as _xxhash

    def _digest(a):
        a = np.ascontiguousarray(a)
        return (a.shape, a.dtype.str, _xxhash.xxh3_64_intdigest(memoryview(a).cast("B")))
except ImportError:                                      # xxhash is optional: the standard library's blake2b instead
    _digest = _digest_stdlib


def _sampled(a):
    """Cheap fingerprint: buffer address, shape and ~4k strided samples + both ends (residency mode 'fast')."""
    a = np.asarray(a)
    f = a.reshape(-1)
    step = max(1, f.size // 4096)
    return (a.__array_interface__["data"][0], a.shape, a.dtype.str, f[::step].tobytes(), f[:64].tobytes(), f[-64:].tobytes())


class _Snapshot:
    """Device copy of a past state, shared by the DeviceArrays of that generation; freed with the last of them."""

    def __init__(self, prop, users):
        self.prop = prop
        self.handle = prop.snapshot()
        self.users = [weakref.ref(a) for a in users]

    def fetch(self, name):
        return self.prop.snapshot_download(self.handle, name)

    def materialize_users(self):
        """Before the context goes away: copy what is still unread to the host."""
        for w in self.users:
            a = w()
            if a is not None and a._host is None and a._snap is self:
                a._get()

    def __del__(self):
        try:
            if self.handle is not None:
                self.prop.snapshot_free(self.handle)
        except Exception:
            pass


class DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    """A slot of the state `RK3` returns.

    An evolving slot (dens, rr, mm, uu, vv; lam, phi, kk, ll with HPROP on) stands for a float64 array that lives in
    HBM and is copied to the host on first access (indexing, arithmetic, `np.asarray`, assignment into another array
    ...); a slot the step did not change holds a private copy of the input.  A loop that only feeds the state back
    into `RK3` never copies or hashes anything: the object is recognised, and its host copy is READ-ONLY unless it is
    written through this object (`a[k] = v`, `a += 1`), which marks it for upload.  (`np.asarray(a)` therefore gives a
    read-only array: copy it to modify it.)  A state that is advanced before it was read stays readable (the library
    keeps a device snapshot of it as long as one of its DeviceArrays is alive).  `lprop.set_lazy_download(False)`
    makes `RK3` return plain, writable ndarrays.
    """
    __array_priority__ = 1000

    def __init__(self, backend, name, shape, gen, host=None):
        self._backend, self._name, self._shape, self._gen = backend, name, tuple(shape), gen
        self._kind = "col" if name in ("uu", "vv") else "rays"
        self._host = None                                   # the materialised ndarray (read-only until written through us)
        self._dirty = False                                 # written through this object: the device no longer matches
        self._snap = None                                   # _Snapshot once the device has moved on
        if host is not None:                                # a slot the step did not change: a private copy of the input
            self._host = np.array(host, dtype=np.float64, copy=True)
            self._host.setflags(write=False)

    # -- materialisation -----------------------------------------------------
    def _get(self):
        if self._host is None:
            self._host = self._backend.fetch(self)
            self._host.setflags(write=False)                # edits must go through this object (see _writable)
            self._snap = None
        return self._host

    def _writable(self):
        """The host copy for an in-place edit (`a[k] = v`, `a += 1`): from now on the device no longer matches, and
        the next hot-path call that gets this object uploads it."""
        h = self._get()
        if not self._dirty:
            h.setflags(write=True)
            self._dirty = True
        return h

    def _pristine(self):
        """True when the values on the device are still what this object shows to the caller (O(1): the host copy is
        read-only unless it was written through this object)."""
        return not self._dirty

    # -- ndarray protocol ------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        a = self._get()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        conv = lambda x: x._get() if isinstance(x, DeviceArray) else x
        outs = kwargs.get("out")
        if outs is not None:
            kwargs["out"] = tuple(x._writable() if isinstance(x, DeviceArray) else x for x in outs)
        res = getattr(ufunc, method)(*[conv(x) for x in inputs], **kwargs)
        if outs is not None and any(isinstance(x, DeviceArray) for x in outs):     # `a += 1` must stay this object
            return outs[0] if len(outs) == 1 else tuple(outs)
        return res

    def __array_function__(self, func, types, args, kwargs):
        if func in (np.shape, np.ndim, np.size) and len(args) == 1 and not kwargs:   # metadata: nothing to copy
            return {np.shape: self.shape, np.ndim: self.ndim, np.size: self.size}[func]
        def conv(x):
            if isinstance(x, DeviceArray):
                return x._get()
            if isinstance(x, (list, tuple)):
                return type(x)(conv(y) for y in x)
            return x
        return func(*[conv(x) for x in args], **{k: conv(v) for k, v in kwargs.items()})

    shape = property(lambda self: self._shape)
    dtype = property(lambda self: np.dtype(np.float64))
    ndim = property(lambda self: len(self._shape))
    size = property(lambda self: int(np.prod(self._shape)))

    def __len__(self):
        return self._shape[0]

    def __iter__(self):
        return iter(self._get())

    def __getitem__(self, k):
        return self._get()[k]

    def __setitem__(self, k, v):
        self._writable()[k] = v

    def __getattr__(self, name):                            # everything else: the ndarray's own attribute
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._get(), name)

    def __repr__(self):
        return repr(self._get()) if self._host is not None else f"DeviceArray({self._name}, shape={self._shape}, on device)"


_SLOT_NAMES = ("dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm", "uu", "vv")
_EVOLVING = {0: "dens", 3: "rr", 7: "mm", 9: "uu", 10: "vv"}
_EVOLVING_HPROP = {1: "lam", 2: "phi", 5: "kk", 6: "ll"}
_EVOLVING_NZ = {4: "drr", 8: "dmm"}                         # with a bvf column (extension)


def _bvf_column():
    """The bvf column when model_config['bvf'] is one (extension), else None."""
    b = model_config['bvf']
    return None if np.ndim(b) == 0 else np.ascontiguousarray(b, dtype=np.float64)


class _Backend:
    """One lazily created GPU context + what is resident in it."""

    def __init__(self):
        self.prop = None
        self.device = 0
        self.cfg = None         # (bvf, phi0, kappa, saturate_online, hprop)
        self.col = None         # copies of grid, grids, rhobar, pressure_gradient
        self.rays = None        # what the device's ray state corresponds to: {slot: key}, see _slot_key
        self.col_uv = None      # the same for the wind columns: (key_uu, key_vv)
        self.gen = {"rays": 0, "col": 0}   # generations of the resident ray state / wind columns (advanced by
                                           # every RK3 and every upload)
        self.live = []          # weak references to the DeviceArrays of the current generations
        self.snaps = []         # weak references to the device snapshots of older generations

    def context(self, ngrid, nray):
        p = self.prop
        if p is None or p.ngrid != ngrid or nray > p.cap:
            cap = max(int(nray), 1024)
            if p is not None and p.ngrid == ngrid:
                cap = max(cap, 2 * p.cap)
            if p is not None:
                self.reset(keep_device=True)
            self.prop = _capi.Propagator(ngrid, cap, device=self.device)
        return self.prop

    def invalidate(self):
        """Forget what is resident (after an error, or when the context goes away): the next call uploads."""
        self.cfg = self.col = self.rays = self.col_uv = None
        self.gen = {k: v + 1 for k, v in self.gen.items()}

    def materialize_live(self):
        """Copy the not yet read results of the current generation to the host (before the device state is
        overwritten by an upload, or the context is closed)."""
        for w in self.live:
            a = w()
            if a is not None and a._host is None:
                a._get()
        self.live = []

    def protect_live(self):
        """Before the device state advances: results of the current generation that nobody has read yet stay
        readable through ONE device snapshot shared by them (freed when the last of them dies or is read)."""
        alive = [a for a in (w() for w in self.live) if a is not None and a._host is None and a._snap is None]
        if alive:
            snap = _Snapshot(self.prop, alive)
            for a in alive:
                a._snap = snap
            self.snaps = [w for w in self.snaps if w() is not None] + [weakref.ref(snap)]
        self.live = []

    def fetch(self, a):
        if a._snap is not None:
            return a._snap.fetch(a._name)
        if a._gen != self.gen[a._kind] or self.prop is None:
            raise RuntimeError("this state is no longer on the device (the GPU context was released)")
        p = self.prop
        if a._name in ("dens", "rr", "mm"):
            return p.download_rays((a._name,))[0]
        if a._name in ("uu", "vv"):
            return p.download_column((a._name,))[0]
        if a._name in ("drr", "dmm"):
            return p.download_extents(which=(a._name,))[0]
        return dict(zip(("lam", "phi", "kk", "ll"), p.download_hprop()))[a._name]

    def reset(self, keep_device=False):
        if self.prop is not None:
            try:
                self.materialize_live()
                for w in self.snaps:
                    if w() is not None:
                        w().materialize_users()
            except Exception:
                pass
            self.prop.close()
        dev = self.device
        self.__init__()
        if keep_device:
            self.device = dev


_backend = _Backend()
_lazy = True
_residency = "safe"


def set_device(device):
    """Select the HIP device of this process (one process per GPU)."""
    _backend.reset()
    _backend.device = int(device)


def release_device():
    """Free the GPU context (the next hot-path call recreates it).  Results that were not read yet are copied to
    the host first."""
    _backend.reset(keep_device=True)


def set_lazy_download(on=True):
    """True (default): `RK3` returns its evolving slots as DeviceArray objects that are copied to the host on first
    access.  False: plain float64 ndarrays, copied eagerly, exactly the reference's return types."""
    global _lazy
    _lazy = bool(on)


def set_residency(mode="safe"):
    """How `RK3` / `rhs_default` decide that a slot the caller passes is already on the device:
      'safe' (default)  the same object as last time AND an unchanged content digest (an in-place edit is seen and
                        uploaded; costs one pass over every writable host array per call, ~0.3 ms per 1e6 rays and
                        array -- the slots `RK3` returned and read-only arrays (`a.setflags(write=False)`, e.g. the
                        statics) are recognised by identity alone);
      'fast'            the same object and a strided sample of ~4k values (an edit of a few rays can go unnoticed);
      'off'             nothing is assumed resident: every call uploads every slot (the reference's semantics at
                        the reference's cost model)."""
    global _residency
    if mode not in ("safe", "fast", "off"):
        raise ValueError("residency mode must be 'safe', 'fast' or 'off'")
    _residency = mode
    _backend.rays = _backend.col_uv = None


def _check_scope():
    if grid is None or grids is None:
        raise RuntimeError("lprop.grid / lprop.grids are not set (raytracer.py:76-77)")
    b = model_config['bvf']
    if np.ndim(b) != 0:                                          # EXTENSION: N as a column on grids
        if np.shape(b) != np.shape(grids):
            raise ValueError("model_config['bvf'] must be a scalar (as in the reference) or an array on lprop.grids")


def _same(a, b):
    return a is b or (a is not None and b is not None and np.array_equal(a, b))


def _slot_key(x):
    """What is remembered about an array that was uploaded / returned, to recognise it next time (weakly: the
    caller's arrays are not kept alive)."""
    if isinstance(x, DeviceArray):
        return ("dev", weakref.ref(x))
    if _residency == "off" or not isinstance(x, np.ndarray):
        return None                                          # lists, scalars ...: converted anew on every call
    if _frozen(x):
        return ("host", weakref.ref(x), None)                # cannot be edited in place: the object is its own key
    return ("host", weakref.ref(x), _digest(x) if _residency == "safe" else _sampled(x))


def _frozen(x):
    """An ndarray nobody can write to in place (read-only, and not a view of something writable): recognised by
    identity alone, in O(1).  `a.setflags(write=False)` on the statics makes the fed-back loop hash nothing."""
    if x.flags.writeable:
        return False
    b = x.base
    while b is not None:
        if not isinstance(b, np.ndarray) or b.flags.writeable:
            return False
        b = b.base
    return True


def _slot_resident(x, key):
    """Is `x` what the device holds for this slot?  The same object as last time and, for host arrays, an unchanged
    content fingerprint (an in-place edit must be uploaded); for a DeviceArray: still of the current generation and
    not edited since it was read."""
    if key is None or _residency == "off" or x is not key[1]():
        return False
    if key[0] == "dev":
        return x._gen == _backend.gen[x._kind] and x._pristine()
    if key[2] is None:
        return _frozen(x)
    return (_digest(x) if _residency == "safe" else _sampled(x)) == key[2]


def _sync_config_and_column(p, uu, vv, force_uv):
    col = _bvf_column()
    cfg = (float(model_config['bvf']) if col is None else col.tobytes(), float(model_config['phi0']),
           float(model_config['kappa']), bool(model_config['saturate_online']), bool(HPROP_GLOBAL))
    if _backend.cfg != cfg:
        _backend.materialize_live()
        p.set_config(float(np.mean(col)) if col is not None else cfg[0], cfg[1], cfg[2], cfg[3], hprop=cfg[4])
        p.set_bvf_column(col)                                    # (None: the scalar of set_config)
        _backend.cfg = cfg
        _backend.rays = None                                     # lam, phi must be (re)uploaded with the rays
    col = _backend.col
    new = (np.asarray(grid, dtype=np.float64), np.asarray(grids, dtype=np.float64),
           np.asarray(rhobar, dtype=np.float64), np.asarray(pressure_gradient, dtype=np.float64))
    if np.shape(new[3]) != (2, len(new[1])):
        raise RuntimeError("lprop.pressure_gradient is not set: call set_pressure_gradient(uu, vv) "
                           "(raytracer.py:99)")
    stale = col is None or any(not _same(a, b) for a, b in zip(col, new))
    keys = _backend.col_uv
    uv_resident = (not force_uv and not stale and keys is not None
                   and _slot_resident(uu, keys[0]) and _slot_resident(vv, keys[1]))
    if not uv_resident:
        _backend.materialize_live()                              # the upload overwrites what they stand for
        p.set_column(new[0], new[1], new[2], new[3], np.asarray(uu, dtype=np.float64), np.asarray(vv, dtype=np.float64))
        _backend.col = tuple(a.copy() for a in new)
        _backend.gen["col"] += 1
        for a in (uu, vv):                                       # results of an earlier call, uploaded again as they are
            if isinstance(a, DeviceArray):
                a._gen = _backend.gen["col"]
        _backend.col_uv = (_slot_key(uu), _slot_key(vv))


_RAY_SLOTS = (0, 3, 4, 5, 6, 7, 8, 2)      # dens rr drr kk ll mm dmm phi
_STATICS = ('dkk', 'dll', 'rr_mm_area')


def _sync_rays(p, var):
    st = [statics[k] for k in _STATICS]                              # KeyError as in :585-587
    res = _backend.rays
    slots = _RAY_SLOTS + ((1,) if HPROP_GLOBAL else ())          # lam matters only with horizontal propagation
    resident = (res is not None and all(_slot_resident(var[i], res.get(i)) for i in slots)
                and all(_slot_resident(a, res.get(k)) for a, k in zip(st, _STATICS)))
    if not resident:
        _backend.materialize_live()                              # the upload overwrites what they stand for
        dens, lam, phi, rr, drr, kk, ll, mm, dmm = [np.asarray(var[i], dtype=np.float64) for i in range(9)]
        p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, st[0], st[1], st[2])
        if HPROP_GLOBAL:
            p.upload_hprop(lam, phi)
        _backend.gen["rays"] += 1
        for i in slots:                                          # results of an earlier call, uploaded again as they are
            if isinstance(var[i], DeviceArray):
                var[i]._gen = _backend.gen["rays"]
        _backend.rays = {i: _slot_key(var[i]) for i in slots}
        _backend.rays.update({k: _slot_key(a) for a, k in zip(st, _STATICS)})
    return not resident


def _sync_config_only(p):
    col = _bvf_column()
    cfg = (float(model_config['bvf']) if col is None else col.tobytes(), float(model_config['phi0']),
           float(model_config['kappa']), bool(model_config['saturate_online']), bool(HPROP_GLOBAL))
    if _backend.cfg != cfg:
        _backend.materialize_live()
        p.set_config(float(np.mean(col)) if col is not None else cfg[0], cfg[1], cfg[2], cfg[3], hprop=cfg[4])
        p.set_bvf_column(col)                                    # (None: the scalar of set_config)
        _backend.cfg = cfg
        _backend.rays = None


def _no_rays(dt, var, flags, tendencies):
    """Zero rays (the reference's numpy code takes empty arrays in its stride: no deposit, the mean flow moves by
    Coriolis force and pressure gradient alone).  The device library wants at least one ray, so one inert ray
    (dens = 0) goes through the same kernels; the ray slots come back empty."""
    _backend.materialize_live()
    p = _backend.context(len(grid), 1)
    _sync_config_and_column(p, var[9], var[10], force_uv=True)
    one = np.ones(1)
    p.upload_rays(0.0 * one, one, one, 1e-4 * one, 0.0 * one, -1e-3 * one, 1e-4 * one, float(model_config['phi0']) * one,
                  1e-4 * one, 1e-4 * one, one)
    if HPROP_GLOBAL:
        p.upload_hprop(0.0 * one, float(model_config['phi0']) * one)
    _backend.rays = _backend.col_uv = None                        # nothing of the caller's is resident
    _backend.gen = {k: v + 1 for k, v in _backend.gen.items()}
    empty = lambda: np.zeros(0)
    if tendencies:
        t = p.rhs(dt, flags)
        return _pack([empty() for _ in range(9)] + [t['uu'], t['vv']])
    p.step(dt, 1, flags | _capi.NO_GRAPH)
    if flags & _capi.FIXED_BACKGROUND:
        return _pack([var[i] for i in range(11)])
    uu, vv = p.download_column()
    return _pack([var[i] for i in range(9)] + [uu, vv])


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


def _guard(fn):
    """Any error of the HIP library leaves the device state unknown: forget what was resident, so that the next call
    uploads again (and, after a time-out of the persistent kernel, proceeds on the per-stage kernels)."""
    def wrapped(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except _capi.MsgwError:
            _backend.invalidate()
            raise
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


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


@_guard
def _rhs(dt, var_in, flags):
    if len(var_in) == 11 and len(var_in[0]) == 0:
        _check_scope()
        return _no_rays(dt, var_in, flags, True)
    p = _prepare(var_in)
    t = p.rhs(dt, flags)
    z = lambda: np.zeros(np.shape(var_in[5]))
    ddrr, ddmm = z(), z()
    lam, phi, kk, ll = z(), z(), z(), z()
    if _bvf_column() is not None:                                # N(z) column: ddrr_st, ddmm_st != 0 (:641, :645)
        ddrr, ddmm = p.download_extents(tendencies=True)
    if HPROP_GLOBAL:                                             # lam, phi, kk, ll have tendencies too (:638-643)
        lam, phi, kk, ll = p.download_hprop(tendencies=True)
    return _pack([t['dens'], lam, phi, t['rr'], ddrr, kk, ll, t['mm'], ddmm, t['uu'], t['vv']])


def RK3(dt, var):
    """lib/libprop.py:680-700: one low-storage RK3 step of the full state (rays AND mean flow: uu, vv advance at
    every stage).

    `model_config['rhs']` is honoured as in the reference (:691).  With one of this module's built-ins
    (`rhs_default`, `rhs_fixed_background`) the whole step runs on the GPU and the state stays resident between
    calls when the caller passes back what it was given.  Any other callable -- e.g. a hook that post-processes
    `lprop.rhs_default(dt, var)` -- is called three times per step by the reference's own six RK lines (:693-698)
    evaluated here on the host; the hook's calls of `rhs_default` still run on the GPU."""
    rhs_ = model_config['rhs']
    if rhs_ is rhs_default:
        return _rk3_device(dt, var, 0)
    if rhs_ is rhs_fixed_background:
        return _rk3_device(dt, var, _capi.FIXED_BACKGROUND)
    if not callable(rhs_):
        raise TypeError("model_config['rhs'] must be callable: rhs(dt, var_in) -> 11 tendencies (lib/libprop.py:691)")
    var = _pack([np.asarray(a, dtype=np.float64) for a in var])     # object array of 11 float64 arrays, slot by slot
    qq = dt * _as_state(rhs_(dt, var))                               # :693
    var = var + qq / 3                                               # :694
    qq = dt * _as_state(rhs_(dt, var)) - 5 / 9 * qq                  # :695
    var = var + 15 / 16 * qq                                         # :696
    qq = dt * _as_state(rhs_(dt, var)) - 153 / 128 * qq              # :697
    var = var + 8 / 15 * qq                                          # :698
    return var


def _as_state(t):
    """The 11 tendencies a user hook returned, as an object array of float64 arrays."""
    if len(t) != 11:
        raise ValueError("the rhs hook must return 11 tendencies (lib/libprop.py:668-674)")
    return _pack([np.asarray(a, dtype=np.float64) for a in t])


@_guard
def _rk3_device(dt, var, flags):
    if len(var) == 11 and len(var[0]) == 0:
        _check_scope()
        return _no_rays(dt, var, flags, False)
    p = _prepare(var)
    _backend.protect_live()                                      # unread results of the state that is about to move
    p.step(dt, 1, flags | _capi.NO_GRAPH)
    moved = dict(_EVOLVING)                                      # slots whose device values have just changed
    if HPROP_GLOBAL:
        moved.update(_EVOLVING_HPROP)
    if _bvf_column() is not None:
        moved.update(_EVOLVING_NZ)
    if flags & _capi.FIXED_BACKGROUND:                           # the hook zeroes the tendencies of slots 9, 10
        moved.pop(9), moved.pop(10)
    _backend.gen["rays"] += 1
    if 9 in moved:
        _backend.gen["col"] += 1
    out = []
    for i in range(11):
        if i in moved:
            out.append(DeviceArray(_backend, moved[i], np.shape(var[i]), _backend.gen["col" if i >= 9 else "rays"]))
        elif isinstance(var[i], DeviceArray):                    # unchanged, a result of an earlier call: still current
            out.append(var[i])
            var[i]._gen = _backend.gen[var[i]._kind]
        elif _lazy and isinstance(var[i], np.ndarray):           # unchanged: a read-only copy that is recognised in O(1)
            out.append(DeviceArray(_backend, _SLOT_NAMES[i], np.shape(var[i]), _backend.gen["col" if i >= 9 else "rays"],
                                   host=var[i]))                 # when it comes back (the caller's array would need a digest)
        else:
            out.append(var[i])
    if not _lazy:                                                # plain, writable ndarrays, copied now
        out = [a._writable() if isinstance(a, DeviceArray) else a for a in out]
    _backend.live = [weakref.ref(a) for a in out if isinstance(a, DeviceArray) and a._host is None]
    # what is on the device now corresponds to `out`
    for i in range(9):
        if i in _backend.rays and out[i] is not var[i]:
            _backend.rays[i] = _slot_key(out[i])
    if out[9] is not var[9] or out[10] is not var[10]:
        _backend.col_uv = (_slot_key(out[9]), _slot_key(out[10]))
    return _pack(out)


def saturation(dt, dens, rr_center, rr_center_st, drr, drr_st, kk, ll, mm_center, mm_center_st,
               direct=False):
    """lib/libprop.py:561-615 on the GPU (the driver's post-step call,
    raytracer.py:182-188, uses direct=True)."""
    _check_scope()
    n = len(dens)
    p = _backend.context(len(grid), n)
    if _backend.col is None:                                     # the caller never ran RK3: any wind column will do,
        z = np.zeros(len(grids))                                 # saturation only reads grids and rhobar
        _sync_config_and_column(p, z, z, force_uv=True)
    else:
        _sync_config_only(p)
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
    bvf = model_config['bvf']
    if np.ndim(bvf) != 0:                                        # EXTENSION: N(z) column, N at .5 * (rr_low + rr_up)
        _check_scope()
        if _backend.col is None:                                 # the caller never ran RK3: any wind column will do
            z = np.zeros(len(grids))
            _sync_config_and_column(p, z, z, force_uv=True)
        else:
            _sync_config_only(p)
        bvf = np.nan                                             # msgw_project_arrays: "the context's column"
    return p.project_arrays(var, bvf, dens, phi, rr_low, rr_up, kk, ll, mm_low, mm_up,
                            dkk, dll, dmm, grid)


# ----------------------------------------------------------------------------
# defaults, as at import of the reference (lib/libprop.py:703-726)
# ----------------------------------------------------------------------------
set_model_setup(
    u0=80, phi0=np.deg2rad(-60), sig_phi=np.deg2rad(3), rr0=30000, rr1=40000, sig_rr=10000, drr=1,
    bvf=0.01, rhs=rhs_default, geostrophy=True, boussinesq=False, hh=8500, rhobar0=1.2, kappa=0.95,
    saturate_online=True)
set_statics(int_dll=1, int_dkk=1, rr_mm_area=0)
