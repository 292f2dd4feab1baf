"""Helpers for the -m gpu tests: build a Propagator (C-ABI context) from a
golden fixture / an oracle Setup + state."""
import numpy as np

from msgwam_amd import _capi


def make_prop(setup, state, cap=None, dtype="f64"):
    dens, lam, phi, rr, drr, kk, ll, mm, dmm, uu, vv = state
    p = _capi.Propagator(len(setup.grid), cap or len(dens), dtype=dtype)
    p.set_config(setup.bvf, setup.phi0, setup.kappa, setup.saturate_online)
    p.set_column(setup.grid, setup.grids, setup.rhobar, setup.pressure_gradient, uu, vv)
    p.upload_rays(dens, rr, drr, kk, ll, mm, dmm, phi, setup.dkk, setup.dll, setup.rr_mm_area)
    return p


def gpu_state(p, state):
    """11-slot list with the evolving slots replaced by the device values."""
    dens, rr, mm = p.download_rays()
    uu, vv = p.download_column()
    out = [np.asarray(s, dtype=np.float64).copy() for s in state]
    out[0], out[3], out[7], out[9], out[10] = dens, rr, mm, uu, vv
    return out
