"""msgwam_amd -- MI355X-native (gfx950) ray-propagation hot path of python-msgwam.

    import msgwam_amd.libprop as lprop     # drop-in for the reference's lib/libprop.py
    from msgwam_amd import Propagator      # resident-state API (state lives in HBM)
"""
from ._capi import Propagator, MsgwError, load_library  # noqa: F401

__all__ = ["Propagator", "MsgwError", "load_library"]
