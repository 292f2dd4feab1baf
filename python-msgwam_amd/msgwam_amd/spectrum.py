"""
Synthetic Gaussian source spectrum -- host-side numpy setup of
the initial ray volumes, the large-N generalisation of the driver's initial
condition (`raytracer.py:71-117`): the same launch layer 0-15 km and wave-action
formula (`raytracer.py:112-117`), tiled deterministically (no RNG) in height,
vertical wavenumber and azimuth so that every box builds the identical spectrum.

Ray order is z-major, then azimuth, then vertical wavenumber:
index = (iz * Nd + id) * Nm + im.  Neighbouring rays (the lanes of a wavefront)
share the launch height and the propagation direction and differ only slightly
in m, so they keep sharing a flux level as they propagate (rays of opposite
azimuth get opposite-signed dm/dt and drift apart, so azimuth must not be the
fastest index); a contiguous index range is a shard.
"""
import numpy as np

ROT_EARTH = 7.2921e-5          # lib/libprop.py:4


def spectrum_shape(nray, nz=100, nd=4):
    """Factor nray = nz * nm * nd (nm = vertical-wavenumber bins)."""
    if nray % (nz * nd):
        raise ValueError(f"nray={nray} must be a multiple of nz*nd={nz * nd}")
    return nz, nray // (nz * nd), nd


def gaussian_spectrum(nray, grids, rhobar, alpha=0.01, bvf=0.01, phi0=0.0,
                      nz=100, nd=4, start=0, stop=None,
                      z_min=0.0, z_max=15e3, z0=7.5e3, sig_z=2e3,
                      lambda_h=50e3, lambda_z=5e3, rel_sig_m=0.1,
                      dkk=1e-4, dll=1e-4):
    """Return the slice [start, stop) of the nray-ray spectrum as a dict of
    float64 arrays: dens, lam, phi, rr, drr, kk, ll, mm, dmm, dkk, dll, area."""
    nz, nm, nd = spectrum_shape(nray, nz, nd)
    stop = nray if stop is None else stop
    idx = np.arange(start, stop, dtype=np.int64)
    i_m = idx % nm
    i_d = (idx // nm) % nd
    i_z = idx // (nd * nm)

    dz_ray = (z_max - z_min) / nz                      # cf. raytracer.py:88-90
    rr = z_min + (i_z + 0.5) * dz_ray
    drr = np.full(idx.shape, dz_ray)

    m0 = -2 * np.pi / lambda_z                         # raytracer.py:85
    sig_m = rel_sig_m * abs(m0)
    dmm_bin = 6 * sig_m / nm
    mm = (m0 - 3 * sig_m) + (i_m + 0.5) * dmm_bin
    dmm = np.full(idx.shape, dmm_bin)

    theta = 2 * np.pi * (i_d + 0.5) / nd
    kh = 2 * np.pi / lambda_h                          # raytracer.py:71
    kk = kh * np.sin(theta)                            # raytracer.py:83-84
    ll = kh * np.cos(theta)

    n = idx.shape[0]
    a_dkk = np.full(n, dkk)
    a_dll = np.full(n, dll)
    area = drr * dmm

    f0 = 2 * ROT_EARTH * np.sin(phi0)
    rhobar_ray = np.interp(rr, grids, rhobar)          # raytracer.py:113
    omh = np.sqrt((bvf ** 2 * (kk ** 2 + ll ** 2) + f0 ** 2 * mm ** 2)
                  / (kk ** 2 + ll ** 2 + mm ** 2))     # lib/libprop.py:383
    amplitude = alpha ** 2 * rhobar_ray / 2 * omh / mm ** 2 / (omh ** 2 - f0 ** 2) * bvf ** 2
    profile = (np.exp(-(rr - z0) ** 2 / 2 / sig_z ** 2)
               * np.exp(-(mm - m0) ** 2 / 2 / sig_m ** 2))
    dens = amplitude * profile / a_dkk / a_dll / dmm
    return dict(dens=dens, lam=np.zeros(n), phi=np.full(n, float(phi0)), rr=rr, drr=drr,
                kk=kk, ll=ll, mm=mm, dmm=dmm, dkk=a_dkk, dll=a_dll, area=area)
