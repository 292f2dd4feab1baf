"""
ORACLE -- test infrastructure, NOT the product.  ctypes loader for the plain-C
restatement (oracle/msgwam_oracle.c).  Same import rule as msgwam_oracle.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmsgwam_oracle.so")
ROT_EARTH = 7.2921e-5

_dp = C.POINTER(C.c_double)


class _Setup(C.Structure):
    _fields_ = [("ngrid", C.c_int), ("grid", _dp), ("grids", _dp), ("rhobar", _dp), ("pg", _dp),
                ("bvf", C.c_double), ("f0", C.c_double), ("kappa", C.c_double),
                ("saturate_online", C.c_int), ("fixed_background", C.c_int)]


def build(force=False):
    src = os.path.join(HERE, "msgwam_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.orc_step.restype = C.c_int
        _lib.orc_step_nz.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class COracle:
    """Holds contiguous copies of the column + config (keeps them alive)."""

    def __init__(self, setup, fixed_background=False):
        self.grid, self.grids = _c(setup.grid), _c(setup.grids)
        self.rhobar, self.pg = _c(setup.rhobar), _c(setup.pressure_gradient)
        # EXTENSION: bvf as a column on grids -> orc_step_nz / orc_rhs_nz (drr, dmm evolve)
        self.bvfcol = _c(setup.bvf) if np.ndim(setup.bvf) else None
        self.bvf = float("nan") if self.bvfcol is not None else float(setup.bvf)
        self.s = _Setup(len(self.grid), _p(self.grid), _p(self.grids), _p(self.rhobar), _p(self.pg),
                        self.bvf, float(2 * ROT_EARTH * np.sin(setup.phi0)), float(setup.kappa),
                        int(bool(setup.saturate_online)), int(fixed_background))
        self.dkk, self.dll, self.area = _c(setup.dkk), _c(setup.dll), _c(setup.rr_mm_area)

    @staticmethod
    def fray(phi):
        return _c(2 * ROT_EARTH * np.sin(phi))

    def step(self, dt, nsteps, state, direct_sat=0):
        """state = 11-slot list; returns a new list (lam, phi, drr, kk, ll, dmm unchanged)."""
        dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = [_c(a).copy() for a in state]
        fr = self.fray(phi)
        if self.bvfcol is not None:
            if direct_sat:
                raise NotImplementedError("the N(z) column extension has no direct saturation")
            rc = lib().orc_step_nz(C.byref(self.s), _p(self.bvfcol), C.c_double(dt), C.c_int(nsteps),
                                   C.c_int64(len(dens)), _p(dens), _p(rr), _p(drr), _p(kk), _p(ll), _p(mm),
                                   _p(dmm), _p(fr), _p(self.dkk), _p(self.dll), _p(self.area), _p(uu), _p(vv))
            if rc:
                raise MemoryError("orc_step_nz")
            return [dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv]
        rc = lib().orc_step(C.byref(self.s), C.c_double(dt), C.c_int(nsteps), C.c_int(direct_sat),
                            C.c_int64(len(dens)), _p(dens), _p(rr), _p(drr), _p(kk), _p(ll), _p(mm),
                            _p(dmm), _p(fr), _p(self.dkk), _p(self.dll), _p(self.area), _p(uu), _p(vv))
        if rc:
            raise MemoryError("orc_step")
        return [dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv]

    def rhs(self, dt, state):
        dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = [_c(a) for a in state]
        n, nc = len(dens), len(self.grids)
        fr = self.fray(phi)
        sd, sr, sm = np.empty(n), np.empty(n), np.empty(n)
        du, dv, flux = np.empty(nc), np.empty(nc), np.empty((2, nc + 1))
        if self.bvfcol is not None:
            sdr, sdm = np.empty(n), np.empty(n)
            lib().orc_rhs_nz(C.byref(self.s), _p(self.bvfcol), C.c_double(dt), C.c_int64(n), _p(dens), _p(rr),
                             _p(drr), _p(kk), _p(ll), _p(mm), _p(dmm), _p(fr), _p(self.dkk), _p(self.dll),
                             _p(self.area), _p(uu), _p(vv), _p(sd), _p(sr), _p(sdr), _p(sm), _p(sdm), _p(du), _p(dv))
            return dict(dens=sd, rr=sr, drr=sdr, mm=sm, dmm=sdm, uu=du, vv=dv)
        lib().orc_rhs(C.byref(self.s), C.c_double(dt), C.c_int64(n), _p(dens), _p(rr), _p(drr), _p(kk),
                      _p(ll), _p(mm), _p(dmm), _p(fr), _p(self.dkk), _p(self.dll), _p(self.area),
                      _p(uu), _p(vv), _p(sd), _p(sr), _p(sm), _p(du), _p(dv), _p(flux))
        return dict(dens=sd, rr=sr, mm=sm, uu=du, vv=dv, pm_flux=flux)

    def project(self, state, G, var):
        dens, lam, phi, rr, drr, kk, ll, mm, dmm = [_c(a) for a in state[:9]]
        G = _c(G)
        fr = self.fray(phi)
        out = np.empty((2, len(G) - 1)) if var == 0 else np.empty(len(G) - 1)
        lib().orc_project(C.c_int64(len(dens)), _p(dens), _p(rr), _p(drr), _p(kk), _p(ll), _p(mm), _p(dmm),
                          _p(fr), _p(self.dkk), _p(self.dll), _p(G), C.c_int(len(G)),
                          C.c_double(self.bvf), C.c_int(var), _p(out))
        return out
