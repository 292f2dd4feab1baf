"""One rank of the 2-processes-on-ONE-GPU exchange test (started by test_gpu_exchange.py;
not collected by pytest).  argv: case.npz rank nranks uid_hex out.npz"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "python-msgwam_amd"))
import numpy as np                                       # noqa: E402
from msgwam_amd import _capi                             # noqa: E402
from msgwam_amd.sharding import shard_bounds             # noqa: E402


def main():
    case, rank, nranks, uid_hex, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    d = np.load(case)
    n = len(d["dens"])
    lo, hi = shard_bounds(n, nranks, rank)
    sl = slice(lo, hi)
    dtype = str(d["dtype"]) if "dtype" in d.files else "f64"
    p = _capi.Propagator(len(d["grid"]), hi - lo, dtype=dtype)
    p.set_config(float(d["bvf"]), float(d["phi0"]), float(d["kappa"]), bool(d["sat"]))
    p.set_column(d["grid"], d["grids"], d["rhobar"], d["pg"], d["uu"], d["vv"])
    p.upload_rays(d["dens"][sl], d["rr"][sl], d["drr"][sl], d["kk"][sl], d["ll"][sl], d["mm"][sl], d["dmm"][sl],
                  d["phi"][sl], d["dkk"][sl], d["dll"][sl], d["area"][sl])
    if "relaunch_frac" in d.files:
        p.set_relaunch(float(d["relaunch_frac"]))
    p.comm_init(bytes.fromhex(uid_hex), rank, nranks)
    cnt0 = p.counters()
    persist = []
    for nsteps in d["calls"]:
        p.step(float(d["dt"]), int(nsteps), int(d["flags"]) if "flags" in d.files else 0)
        persist.append(p.counters()["persist_steps"])
    dens, rr, mm = p.download_rays()
    uu, vv = p.download_column()
    np.savez(out, dens=dens, rr=rr, mm=mm, uu=uu, vv=vv, lo=lo, hi=hi, exchange=cnt0["exchange"],
             transport=cnt0["transport"], tenants=cnt0["tenants"], persist=np.array(persist))
    p.close()


if __name__ == "__main__":
    main()
