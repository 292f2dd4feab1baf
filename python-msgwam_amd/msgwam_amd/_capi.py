"""
ctypes binding of the C ABI declared in include/msgwam_hip.h (the in-tree
gfx950 shared library libmsgwam_hip.so) and a thin object wrapper.

There is NO CPU fallback: if the library is missing or no MI355X is visible,
everything here raises.  Build with `python __graft_entry__.py` or
`make -C python-msgwam_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSGW_LIBRARY", os.path.join(HERE, "libmsgwam_hip.so"))   # override: diagnostic builds
ROT_EARTH = 7.2921e-5          # lib/libprop.py:4

FIXED_BACKGROUND = 1
DIRECT_SAT_QUIRK = 2
DIRECT_SAT = 4
NO_GRAPH = 8
TIME_KERNELS = 16
RELAUNCH = 32           # extension, see include/msgwam_hip.h
DTYPE_F32 = 1           # msgw_create_ex flag: float32 ray state (BASELINE config 5)
TRANSPORTS = {0: "none", 1: "rccl", 2: "host_shm", 3: "device_ipc"}

EXPORTS = [
    "msgw_abi_version", "msgw_last_error", "msgw_create", "msgw_create_ex", "msgw_destroy", "msgw_set_config",
    "msgw_set_column", "msgw_upload_rays", "msgw_step", "msgw_rhs", "msgw_project",
    "msgw_project_arrays", "msgw_saturation", "msgw_download_rays", "msgw_download_column",
    "msgw_sync", "msgw_comm_unique_id", "msgw_comm_init", "msgw_set_tuning", "msgw_counters", "msgw_set_relaunch", "msgw_set_relaunch_source", "msgw_upload_hprop", "msgw_download_hprop",
    "msgw_snapshot_create", "msgw_snapshot_download", "msgw_snapshot_destroy", "msgw_set_bvf_column", "msgw_download_extents",
    "msgw_probe_arith",
]

_dp = C.POINTER(C.c_double)


class Counters(C.Structure):
    _fields_ = [("last_step_ms", C.c_double), ("ray_kernel_ms_sum", C.c_double),
                ("ray_kernel_launches", C.c_int64), ("ray_steps_total", C.c_int64),
                ("nray", C.c_int64), ("ngrid", C.c_int32), ("blocks", C.c_int32),
                ("graph_steps", C.c_int32), ("nranks", C.c_int32), ("persist_steps", C.c_int32),
                ("exchange", C.c_int32),
                ("persist_resident_tiles", C.c_int32), ("elem_bytes", C.c_int32),
                ("transport", C.c_int32), ("tenants", C.c_int32),
                ("launch_grid", C.c_int32), ("launch_ray_workgroups", C.c_int32), ("launch_reducers", C.c_int32),
                ("fixed_narrow", C.c_int32), ("carried_flux", C.c_int32), ("algorithmic_bytes_total", C.c_double),
                ("cooperative", C.c_int32), ("coop_refused", C.c_int32)]


class MsgwError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen the HIP library; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsgwError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.msgw_last_error.restype = C.c_char_p
    lib.msgw_last_error.argtypes = [C.c_void_p]
    lib.msgw_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int]
    lib.msgw_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_uint]
    lib.msgw_destroy.argtypes = [C.c_void_p]
    lib.msgw_set_config.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.msgw_set_column.argtypes = [C.c_void_p] + [_dp] * 6
    lib.msgw_upload_rays.argtypes = [C.c_void_p, C.c_int64] + [_dp] * 11
    lib.msgw_step.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_uint]
    lib.msgw_rhs.argtypes = [C.c_void_p, C.c_double, C.c_uint] + [_dp] * 6
    lib.msgw_project.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
    lib.msgw_project_arrays.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double] + [_dp] * 11 + [_dp, C.c_int, _dp]
    lib.msgw_saturation.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_int] + [_dp] * 12 + [_dp]
    lib.msgw_download_rays.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp]
    lib.msgw_download_column.argtypes = [C.c_void_p, _dp, _dp]
    lib.msgw_sync.argtypes = [C.c_void_p]
    lib.msgw_comm_unique_id.argtypes = [C.c_void_p]
    lib.msgw_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.msgw_set_tuning.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.msgw_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    lib.msgw_set_relaunch.argtypes = [C.c_void_p, C.c_double]
    lib.msgw_set_relaunch_source.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp]
    lib.msgw_upload_hprop.argtypes = [C.c_void_p, C.c_int64, _dp, _dp]
    lib.msgw_download_hprop.argtypes = [C.c_void_p, C.c_int64, C.c_int, _dp, _dp, _dp, _dp]
    lib.msgw_snapshot_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.msgw_snapshot_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _dp]
    lib.msgw_set_bvf_column.argtypes = [C.c_void_p, _dp]
    lib.msgw_download_extents.argtypes = [C.c_void_p, C.c_int64, C.c_int, _dp, _dp]
    lib.msgw_snapshot_destroy.argtypes = [C.c_void_p, C.c_void_p]
    lib.msgw_probe_arith.argtypes = [C.c_void_p, C.c_int64, _dp, C.c_double, _dp, _dp, _dp, _dp]
    if lib.msgw_abi_version() != 3:
        raise MsgwError("libmsgwam_hip.so has an unexpected ABI version")
    _lib = lib
    return lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def coriolis(phi):
    """f = 2*Omega*sin(phi) exactly as the reference evaluates it (lib/libprop.py:382)."""
    return 2 * ROT_EARTH * np.sin(phi)


def comm_unique_id():
    buf = C.create_string_buffer(128)
    lib = load_library()
    rc = lib.msgw_comm_unique_id(buf)
    if rc:
        raise MsgwError(f"msgw_comm_unique_id: {lib.msgw_last_error(None).decode()}")
    return buf.raw


class Propagator:
    """One GPU context: the ray state and the column live in HBM between calls."""

    def __init__(self, ngrid, nray_cap, device=0, dtype="f64"):
        """dtype "f64" (default; the reference's precision) or "f32" (BASELINE config 5: float32 ray state)."""
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        self.lib = load_library()
        self.ctx = C.c_void_p()
        self.ngrid, self.cap, self.n, self.dtype = int(ngrid), int(nray_cap), 0, dtype
        self._given = {}            # what save_checkpoint needs besides the device state (references, no copies)
        rc = self.lib.msgw_create_ex(C.byref(self.ctx), int(device), self.cap, self.ngrid,
                                     DTYPE_F32 if dtype == "f32" else 0)
        if rc:
            raise MsgwError(f"msgw_create_ex: {self.lib.msgw_last_error(None).decode()} (rc={rc})")

    def _chk(self, rc, what):
        if rc:
            raise MsgwError(f"{what}: {self.lib.msgw_last_error(self.ctx).decode()} (rc={rc})")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.msgw_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------
    def set_config(self, bvf, phi0, kappa, saturate_online, hprop=False):
        self._chk(self.lib.msgw_set_config(self.ctx, float(bvf), float(coriolis(phi0)), float(kappa),
                                           int(bool(saturate_online)), int(bool(hprop))), "msgw_set_config")
        self._given["config"] = np.array([float(bvf), float(phi0), float(kappa), float(bool(saturate_online)),
                                          float(bool(hprop))])

    def set_column(self, grid, grids, rhobar, pressure_gradient, uu, vv):
        a = [_c(x) for x in (grid, grids, rhobar, pressure_gradient, uu, vv)]
        nc = self.ngrid - 1
        if a[0].shape != (self.ngrid,) or a[1].shape != (nc,) or a[2].shape != (nc,) \
                or a[3].shape != (2, nc) or a[4].shape != (nc,) or a[5].shape != (nc,):
            raise ValueError("column arrays do not match ngrid")
        self._chk(self.lib.msgw_set_column(self.ctx, *[_p(x) for x in a]), "msgw_set_column")
        self._given.update(grid=a[0], grids=a[1], rhobar=a[2], pressure_gradient=a[3])

    def upload_rays(self, dens, rr, drr, kk, ll, mm, dmm, phi, dkk, dll, rr_mm_area):
        n = len(dens)
        fray = np.broadcast_to(_c(coriolis(np.asarray(phi, dtype=np.float64))), (n,))
        a = [_c(np.broadcast_to(x, (n,))) for x in (dens, rr, drr, kk, ll, mm, dmm, fray, dkk, dll, rr_mm_area)]
        self._chk(self.lib.msgw_upload_rays(self.ctx, n, *[_p(x) for x in a]), "msgw_upload_rays")
        self.n = n
        self._given.update(drr=a[2], kk=a[3], ll=a[4], dmm=a[6], phi_static=_c(np.broadcast_to(phi, (n,))), dkk=a[8],
                           dll=a[9], rr_mm_area=a[10], src_dens=a[0], src_rr=a[1], src_mm=a[5])

    def set_tuning(self, blocks_per_cu=4, graph_steps=0):
        self._chk(self.lib.msgw_set_tuning(self.ctx, int(blocks_per_cu), int(graph_steps)), "msgw_set_tuning")

    def set_bvf_column(self, bvf):
        """EXTENSION: N as a column on grids (None: back to the scalar of set_config).  Before upload_rays."""
        if bvf is None:
            self._chk(self.lib.msgw_set_bvf_column(self.ctx, None), "msgw_set_bvf_column")
            self._given.pop("bvf_column", None)
            return
        b = _c(bvf)
        if b.shape != (self.ngrid - 1,):
            raise ValueError("the bvf column must have ngrid-1 values (on grids)")
        self._chk(self.lib.msgw_set_bvf_column(self.ctx, _p(b)), "msgw_set_bvf_column")
        self._given["bvf_column"] = b

    def download_extents(self, tendencies=False, which=("drr", "dmm")):
        """Slots 4, 8 (drr, dmm; they evolve only with an N(z) column), or their tendencies after rhs()."""
        out = {k: np.empty(self.n) for k in which}
        self._chk(self.lib.msgw_download_extents(self.ctx, self.n, int(bool(tendencies)), _p(out.get("drr")),
                                                 _p(out.get("dmm"))), "msgw_download_extents")
        return tuple(out[k] for k in which)

    def upload_hprop(self, lam, phi):
        """HPROP on: slots 1, 2 of the rays of the last upload_rays."""
        a = [_c(np.broadcast_to(x, (self.n,))) for x in (lam, phi)]
        self._chk(self.lib.msgw_upload_hprop(self.ctx, self.n, _p(a[0]), _p(a[1])), "msgw_upload_hprop")

    def download_hprop(self, tendencies=False):
        """HPROP on: (lam, phi, kk, ll), or their tendencies after rhs()."""
        out = [np.empty(self.n) for _ in range(4)]
        self._chk(self.lib.msgw_download_hprop(self.ctx, self.n, int(bool(tendencies)), *[_p(x) for x in out]),
                  "msgw_download_hprop")
        return out

    def set_relaunch(self, frac=1e-6):
        """EXTENSION (not in the reference): broken-ray fraction of the RELAUNCH flag."""
        self._chk(self.lib.msgw_set_relaunch(self.ctx, float(frac)), "msgw_set_relaunch")
        self._given["relaunch_frac"] = np.array(float(frac))

    def set_relaunch_source(self, dens, rr, mm):
        """EXTENSION: the (dens, rr, mm) a recycled slot returns to (default: the state upload_rays was given)."""
        a = [_c(np.broadcast_to(x, (self.n,))) for x in (dens, rr, mm)]
        self._chk(self.lib.msgw_set_relaunch_source(self.ctx, self.n, *[_p(x) for x in a]), "msgw_set_relaunch_source")
        self._given.update(src_dens=a[0], src_rr=a[1], src_mm=a[2])

    # -- hot path --------------------------------------------------------------
    def step(self, dt, nsteps=1, flags=0):
        self._chk(self.lib.msgw_step(self.ctx, float(dt), int(nsteps), int(flags)), "msgw_step")

    def rhs(self, dt, flags=0):
        n, nc = self.n, self.ngrid - 1
        out = dict(dens=np.empty(n), rr=np.empty(n), mm=np.empty(n), uu=np.empty(nc), vv=np.empty(nc),
                   pm_flux=np.empty((2, self.ngrid)))
        self._chk(self.lib.msgw_rhs(self.ctx, float(dt), int(flags), _p(out["dens"]), _p(out["rr"]),
                                    _p(out["mm"]), _p(out["uu"]), _p(out["vv"]), _p(out["pm_flux"])), "msgw_rhs")
        return out

    @staticmethod
    def _proj_shape(var, nG):
        """lib/libprop.py:147, :165, :182, :200, :210: cell centres for var 0-2, interfaces for var 3, 4."""
        return {0: (2, nG - 1), 1: (nG - 1,), 2: (nG - 1,), 3: (nG,), 4: (2, nG)}[int(var)]

    def project(self, var, G):
        G = _c(G)
        out = np.empty(self._proj_shape(var, len(G)))
        self._chk(self.lib.msgw_project(self.ctx, int(var), _p(G), len(G), _p(out)), "msgw_project")
        return out

    def project_arrays(self, var, bvf, dens, phi, rr_low, rr_up, kk, ll, mm_low, mm_up, dkk, dll, dmm, G):
        n = len(dens)
        fray = coriolis(np.asarray(phi, dtype=np.float64))
        a = [_c(np.broadcast_to(x, (n,))) for x in (dens, rr_low, rr_up, kk, ll, mm_low, mm_up, dkk, dll, dmm, fray)]
        G = _c(G)
        out = np.empty(self._proj_shape(var, len(G)))
        if n == 0:                                               # nothing to project (the library wants n >= 1)
            out[...] = 0.0
            return out
        self._chk(self.lib.msgw_project_arrays(self.ctx, n, int(var), float(bvf), *[_p(x) for x in a],
                                               _p(G), len(G), _p(out)), "msgw_project_arrays")
        return out

    def saturation(self, dt, direct, dens, rr_center, rr_center_st, drr, drr_st, kk, ll, mm_center,
                   mm_center_st, dkk, dll, rr_mm_area):
        n = len(dens)
        a = [_c(np.broadcast_to(x, (n,))) for x in (dens, rr_center, rr_center_st, drr, drr_st, kk, ll,
                                                    mm_center, mm_center_st, dkk, dll, rr_mm_area)]
        out = np.empty(n)
        if n == 0:
            return out
        self._chk(self.lib.msgw_saturation(self.ctx, n, float(dt), int(bool(direct)), *[_p(x) for x in a],
                                           _p(out)), "msgw_saturation")
        return out

    def download_rays(self, which=("dens", "rr", "mm")):
        """The evolving per-ray slots (blocking); `which` selects a subset, returned in that order."""
        out = {k: np.empty(self.n) for k in which}
        self._chk(self.lib.msgw_download_rays(self.ctx, self.n, _p(out.get("dens")), _p(out.get("rr")),
                                              _p(out.get("mm"))), "msgw_download_rays")
        return tuple(out[k] for k in which)

    def download_column(self, which=("uu", "vv")):
        nc = self.ngrid - 1
        out = {k: np.empty(nc) for k in which}
        self._chk(self.lib.msgw_download_column(self.ctx, _p(out.get("uu")), _p(out.get("vv"))), "msgw_download_column")
        return tuple(out[k] for k in which)

    # -- snapshots (lazy host copies) ------------------------------------------
    _SNAP_SLOTS = ("dens", "rr", "mm", "uu", "vv", "lam", "phi", "kk", "ll", "drr", "dmm")   # MSGW_SLOT_*

    def snapshot(self):
        """Stream-ordered device copy of the evolving slots; returns an opaque handle (release with snapshot_free)."""
        h = C.c_void_p()
        self._chk(self.lib.msgw_snapshot_create(self.ctx, C.byref(h)), "msgw_snapshot_create")
        return (h, self.n)

    def snapshot_download(self, snap, name):
        h, n = snap
        out = np.empty(self.ngrid - 1 if name in ("uu", "vv") else n)
        self._chk(self.lib.msgw_snapshot_download(self.ctx, h, self._SNAP_SLOTS.index(name), _p(out)),
                  "msgw_snapshot_download")
        return out

    def snapshot_free(self, snap):
        if getattr(self, "ctx", None):
            self.lib.msgw_snapshot_destroy(self.ctx, snap[0])

    # -- checkpoint / resume ----------------------------------------------------
    # the reference keeps its whole history in RAM (raytracer.py:125-136) and writes nothing; here the
    # state lives in HBM, so a run that is to survive its process needs a file.  One .npz (numpy's own container,
    # read back with allow_pickle=False): the evolving slots as they are on the device NOW, and the configuration,
    # column and static per-ray arrays this object was given (kept by reference, not copied).
    CHECKPOINT_FORMAT = 1

    def save_checkpoint(self, path, **meta):
        """Write everything `load_checkpoint` needs to continue this run; `meta` (numbers / small arrays, e.g. step=...)
        is stored beside it under 'meta_<name>'.  Blocking (it downloads the state)."""
        g = self._given
        need = ("config", "grid", "drr")
        if any(k not in g for k in need):
            raise MsgwError("save_checkpoint: set_config, set_column and upload_rays come first")
        hprop, nz = bool(g["config"][4]), "bvf_column" in g
        out = {k: g[k] for k in ("config", "grid", "grids", "rhobar", "pressure_gradient", "phi_static", "dkk", "dll",
                                 "rr_mm_area", "src_dens", "src_rr", "src_mm")}
        out["dens"], out["rr"], out["mm"] = self.download_rays()
        out["uu"], out["vv"] = self.download_column()
        out["drr"], out["dmm"], out["kk"], out["ll"] = g["drr"], g["dmm"], g["kk"], g["ll"]
        if nz:
            out["bvf_column"] = g["bvf_column"]
            out["drr"], out["dmm"] = self.download_extents()
        if hprop:
            out["lam"], out["phi"], out["kk"], out["ll"] = self.download_hprop()
        if "relaunch_frac" in g:
            out["relaunch_frac"] = g["relaunch_frac"]
        out["format"] = np.array(self.CHECKPOINT_FORMAT)
        out["ngrid"] = np.array(self.ngrid)
        out["float32_state"] = np.array(self.dtype == "f32")
        for k, v in meta.items():
            out["meta_" + k] = np.asarray(v)
        with open(path, "wb") as f:                            # (np.savez would append ".npz" to a bare name)
            np.savez(f, **out)

    @classmethod
    def load_checkpoint(cls, path, device=0, nray_cap=None):
        """A new context that continues where `save_checkpoint` stopped; returns (propagator, meta dict).
        float64 contexts resume bit for bit (what the next step computes from is exactly what was on the device)."""
        with np.load(path, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
        if int(d["format"]) != cls.CHECKPOINT_FORMAT:
            raise MsgwError(f"checkpoint format {int(d['format'])}, this build reads {cls.CHECKPOINT_FORMAT}")
        n = len(d["dens"])
        p = cls(int(d["ngrid"]), int(nray_cap or n), device=device, dtype="f32" if bool(d["float32_state"]) else "f64")
        try:
            bvf, phi0, kappa, sat, hprop = d["config"]
            p.set_config(bvf, phi0, kappa, bool(sat), hprop=bool(hprop))
            if "bvf_column" in d:
                p.set_bvf_column(d["bvf_column"])
            p.set_column(d["grid"], d["grids"], d["rhobar"], d["pressure_gradient"], d["uu"], d["vv"])
            p.upload_rays(d["dens"], d["rr"], d["drr"], d["kk"], d["ll"], d["mm"], d["dmm"], d["phi_static"], d["dkk"],
                          d["dll"], d["rr_mm_area"])
            if bool(hprop):
                p.upload_hprop(d["lam"], d["phi"])
            p.set_relaunch_source(d["src_dens"], d["src_rr"], d["src_mm"])
            if "relaunch_frac" in d:
                p.set_relaunch(float(d["relaunch_frac"]))
        except Exception:
            p.close()
            raise
        return p, {k[5:]: v for k, v in d.items() if k.startswith("meta_")}

    def probe_arith(self, x, d, y=None):
        """Test support: (sqrt(x), x / d[, x / y]) as the ray kernels evaluate them (include/msgwam_hip.h)."""
        x = _c(x)
        s, q = np.empty(len(x)), np.empty(len(x))
        if y is None:
            self._chk(self.lib.msgw_probe_arith(self.ctx, len(x), _p(x), float(d), _p(s), _p(q), None, None), "msgw_probe_arith")
            return s, q
        y = _c(np.broadcast_to(y, x.shape))
        t = np.empty(len(x))
        self._chk(self.lib.msgw_probe_arith(self.ctx, len(x), _p(x), float(d), _p(s), _p(q), _p(y), _p(t)), "msgw_probe_arith")
        return s, q, t

    def sync(self):
        self._chk(self.lib.msgw_sync(self.ctx), "msgw_sync")

    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.lib.msgw_comm_init(self.ctx, buf, int(rank), int(nranks)), "msgw_comm_init")

    def counters(self):
        c = Counters()
        self._chk(self.lib.msgw_counters(self.ctx, C.byref(c)), "msgw_counters")
        return {k: getattr(c, k) for k, _ in Counters._fields_}
