"""Standalone raytracing entry point, as pyc2ray/raytracing.py:34-108."""
import time

import numpy as np

from . import _capi, _residency
from .asora_core import cuda_is_init
from .load_extensions import load_asora, load_c2ray
from .utils import printlog
from .utils.sourceutils import format_sources

__all__ = ['do_raytracing']


def do_raytracing(dr,
                  src_flux, src_pos,
                  use_gpu, max_subbox, subboxsize, loss_fraction,
                  ndens, xh_av,
                  photo_thin_table, photo_thick_table,
                  heat_thin_table, heat_thick_table,
                  minlogtau, dlogtau,
                  R_max_LLS,
                  sig,
                  logfile="pyC2Ray.log", quiet=False, stats=False):
    """Raytrace all sources once and return the photo-ionisation rate grid.

    Same 17 positional arguments as the reference (pyc2ray/raytracing.py:34-43).  Returns
    ``(phi_ion, phi_heat)``.  On the GPU path the reference's return statement refers to an
    undefined ``phi_heat`` (raytracing.py:108, NameError; GPU heating is a TODO there,
    c2ray_base.py:424-426); here the photo-heating rate is computed on the GPU when non-zero heating
    tables are passed (arithmetic of the Fortran path, photorates.f90:118,124) and is ``None`` otherwise.

    ``use_gpu=False`` selects the semantics of the reference's Fortran CPU raytracer (cubic sub-boxes grown
    until the photon loss is below loss_fraction, raytracing.py:89-95) -- evaluated on the GPU through
    ``libc2ray.raytracing.do_all_sources``: returns ``(phi_ion, phi_heat)`` in Fortran order, or
    ``(phi_ion, nsubbox, photonloss)`` with ``stats=True`` (raytracing.py:105-108).
    """
    if use_gpu and not cuda_is_init():
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
    _residency.reclaim()              # this call overwrites device grids a resident C2Ray object may be relying on

    NumSrc = src_flux.shape[0]
    N = ndens.shape[0]
    NumTau = photo_thin_table.shape[0]

    printlog(f"dr [Mpc]: {dr/3.086e24:.3e}", logfile, quiet)
    printlog(f"Running on {NumSrc:n} source(s), total normalized ionizing flux: {src_flux.sum():.2e}", logfile, quiet)
    if not use_gpu:
        printlog(f"Mean density (cgs): {ndens.mean():.3e}, Mean ionized fraction: {xh_av.mean():.3e}", logfile, quiet)
        trt0 = time.time()
        printlog("Doing Raytracing...", logfile, quiet, ' ')
        phi_ion = np.zeros((N, N, N), order='F')                      # raytracing.py:80-83
        phi_heat = np.zeros((N, N, N), order='F')
        coldensh_out = np.zeros((N, N, N), order='F')
        nsubbox, photonloss = load_c2ray().raytracing.do_all_sources(
            src_flux, src_pos, max_subbox, subboxsize, coldensh_out, sig, dr, np.asfortranarray(ndens), xh_av,
            phi_ion, phi_heat, loss_fraction, photo_thin_table, photo_thick_table, heat_thin_table,
            heat_thick_table, minlogtau, dlogtau, R_max_LLS)
        printlog(f"took {(time.time()-trt0) : .1f} s.", logfile, quiet)
        printlog(f"Average number of subboxes: {nsubbox/NumSrc:n}, Total photon loss: {photonloss:.3e}", logfile, quiet)
        if stats:
            return phi_ion, nsubbox, photonloss
        return phi_ion, phi_heat

    libasora = load_asora()
    srcpos_flat, normflux_flat = format_sources(src_pos, src_flux)
    libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc)
    libasora.grid_to_device(_capi.GRID_NDENS, ndens)
    libasora.grid_to_device(_capi.GRID_XH_AV, xh_av)
    # the reference prints the means before the copies (raytracing.py:72-76); here they are summed on the device
    printlog(f"Mean density (cgs): {libasora.grid_sum(_capi.GRID_NDENS) / N ** 3:.3e}, "
             f"Mean ionized fraction: {libasora.grid_sum(_capi.GRID_XH_AV) / N ** 3:.3e}", logfile, quiet)
    printlog("Copied source data to device.", logfile, quiet)

    # photo-heating: only when real heating tables are passed (the reference's callers pass zeros when
    # compute_heating_rates is off, c2ray_base.py:430-431)
    want_heat = (heat_thin_table is not None and heat_thick_table is not None
                 and (np.any(heat_thin_table) or np.any(heat_thick_table)))
    if want_heat:
        libasora.heat_table_to_device(heat_thin_table, heat_thick_table, NumTau)
    libasora.set_option(_capi.OPT_HEATING, 1 if want_heat else 0)

    trt0 = time.time()
    printlog("Doing Raytracing...", logfile, quiet, ' ')
    try:
        libasora.raytrace_device(R_max_LLS, sig, dr, 0, NumSrc, minlogtau, dlogtau, NumTau)
    finally:
        libasora.set_option(_capi.OPT_HEATING, 0)
    phi_ion = libasora.grid_to_host(_capi.GRID_PHI_ION, libasora.host_empty((N, N, N)))
    phi_heat = libasora.grid_to_host(_capi.GRID_PHI_HEAT, libasora.host_empty((N, N, N))) if want_heat else None
    printlog(f"took {(time.time()-trt0) : .1f} s.", logfile, quiet)
    return phi_ion, phi_heat
