"""Shared helpers for the parity tests: load a golden fixture and turn it into
oracle `Setup` + state lists."""
import os

import numpy as np

from oracle import msgwam_oracle as orc

STATE_KEYS = ["dens", "lam", "phi", "rr", "drr", "kk", "ll", "mm", "dmm", "uu", "vv"]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def setup_from(d):
    s = orc.Setup(d["grid"], bvf=float(d["bvf"]), phi0=float(d["phi0"]), kappa=float(d["kappa"]),
                  saturate_online=bool(int(d["saturate_online"])),
                  dkk=d["dkk"], dll=d["dll"], rr_mm_area=d["area"])
    # the fixture carries the reference's own rhobar / pressure gradient
    np.testing.assert_array_equal(s.rhobar, d["rhobar"])
    s.pressure_gradient = d["pg"].copy()
    return s


def state_from(d, prefix):
    return [d[f"{prefix}_{k}"].copy() for k in STATE_KEYS]


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), np.finfo(np.float64).tiny)
    return float(np.max(np.abs(a - b) / scale)) if a.size else 0.0
