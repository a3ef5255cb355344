"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE, see oracle/__init__.py).

Function names mirror the reference entry points they restate:
  do_all_sources        <- src/c2ray/raytracing.f90:52   (Fortran CPU path, cubic traversal)
  asora_do_all_sources  <- src/asora/raytracing.cu:79    (GPU path semantics, shell traversal)
  global_pass           <- src/c2ray/chemistry.f90:13
  doric / do_chemistry  <- src/c2ray/chemistry.f90:221 / :117
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

ASORA_CONSTS = 1
THIN_TAU_OUT = 2
GREY = 4
PER_SOURCE_FLUX = 8
#: flags that make the oracle follow the CUDA (libasora) semantics instead of the Fortran
ASORA_MODE = ASORA_CONSTS | THIN_TAU_OUT | PER_SOURCE_FLUX

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build():
    """(Re)build liboracle.so (and oracle/_ref when the reference checkout is present)."""
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if (not os.path.exists(_LIB)
                or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "c2ray_oracle.c"))):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_do_chemistry.restype = C.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _opt(a):
    return None if a is None else _d(a)


def do_all_sources(normflux, srcpos, max_subbox, subboxsize, sig, dr, ndens, xh_av,
                   loss_fraction, thin, thick, minlogtau, dlogtau, R_max_LLS,
                   heat_thin=None, heat_thick=None, NumTau=None, flags=0):
    """Fortran-path raytrace. ndens/xh_av: (N,N,N) any order (logical [i,j,k]);
    srcpos: (3,Ns) 1-based.  Returns dict(phi_ion, phi_heat, coldens, nsubbox, photon_loss),
    arrays Fortran-ordered, coldens = scratch of the LAST source."""
    N = ndens.shape[0]
    nd = np.asfortranarray(ndens, dtype=np.float64)
    xh = np.asfortranarray(xh_av, dtype=np.float64)
    flux = np.ascontiguousarray(normflux, dtype=np.float64)
    pos = np.asfortranarray(np.asarray(srcpos).astype(np.int32))
    assert pos.shape == (3, flux.shape[0])
    thin = np.ascontiguousarray(thin, dtype=np.float64)
    thick = np.ascontiguousarray(thick, dtype=np.float64)
    ht = None if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
    hk = None if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
    if NumTau is None:
        NumTau = thin.shape[0]
    phi = np.zeros((N, N, N), order="F")
    heat = np.zeros((N, N, N), order="F")
    cd = np.zeros((N, N, N), order="F")
    nbox = C.c_int(0)
    loss = C.c_double(0.0)
    lib().oracle_do_all_sources(
        _d(flux), pos.ctypes.data_as(_ip), C.c_int(max_subbox), C.c_int(subboxsize), _d(cd),
        C.c_double(sig), C.c_double(dr), _d(nd), _d(xh), _d(phi), _d(heat), C.c_float(loss_fraction),
        _d(thin), _d(thick), _opt(ht), _opt(hk), C.c_double(minlogtau), C.c_double(dlogtau),
        C.c_double(R_max_LLS), C.c_int(NumTau), C.c_int(thin.shape[0]), C.c_int(flux.shape[0]),
        C.c_int(N), C.c_int(N), C.c_int(N), C.c_int(flags), C.byref(nbox), C.byref(loss))
    return dict(phi_ion=phi, phi_heat=heat, coldens=cd, nsubbox=nbox.value, photon_loss=loss.value)


def asora_do_all_sources(R, sig, dr, ndens, xh_av, src_pos0, src_flux, thin, thick,
                         minlogtau, dlogtau, NumTau=None, flags=ASORA_MODE, want_coldens=False):
    """ASORA-semantics raytrace.  ndens/xh_av: (N,N,N) logical [i,j,k]; src_pos0: flat int32
    0-based [x0,y0,z0,x1,...] (format_sources layout).  Returns dict(phi_ion (N,N,N) C-order,
    coldens (last source, if requested), visited)."""
    N = ndens.shape[0]
    nd = np.ascontiguousarray(ndens, dtype=np.float64)
    xh = np.ascontiguousarray(xh_av, dtype=np.float64)
    pos = np.ascontiguousarray(src_pos0, dtype=np.int32)
    flux = np.ascontiguousarray(src_flux, dtype=np.float64)
    thin = np.ascontiguousarray(thin, dtype=np.float64)
    thick = np.ascontiguousarray(thick, dtype=np.float64)
    if NumTau is None:
        NumTau = thin.shape[0]
    phi = np.zeros((N, N, N))
    cd = np.zeros((N, N, N)) if want_coldens else None
    visited = C.c_long(0)
    lib().oracle_asora_do_all_sources(
        C.c_double(R), C.c_double(sig), C.c_double(dr), _d(nd), _d(xh), _d(phi),
        pos.ctypes.data_as(_ip), _d(flux), C.c_int(flux.shape[0]), C.c_int(N), _d(thin), _d(thick),
        C.c_double(minlogtau), C.c_double(dlogtau), C.c_int(NumTau), C.c_int(thin.shape[0]),
        C.c_int(flags), _opt(cd), C.byref(visited))
    return dict(phi_ion=phi, coldens=cd, visited=visited.value)


def global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c):
    """Chemistry pass.  All grids must share shape; results are returned as NEW arrays
    (xh_av_new, xh_intermed_new, conv_flag, total_inner_iterations) in C order of the logical grid."""
    shp = ndens.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64).ravel().copy()
    nd, tp, x0, xa, xi, ph = map(f, (ndens, temp, xh, xh_av, xh_intermed, phi_ion))
    conv = C.c_int(0)
    its = C.c_long(0)
    lib().oracle_global_pass(C.c_double(dt), _d(nd), _d(tp), _d(x0), _d(xa), _d(xi), _d(ph),
                             C.c_double(bh00), C.c_double(albpow), C.c_double(colh0),
                             C.c_double(temph0), C.c_double(abu_c), C.c_long(nd.size),
                             C.byref(conv), C.byref(its))
    return xa.reshape(shp), xi.reshape(shp), conv.value, its.value


def doric(xh_old, dt, temp, rhe, phi, bh00, albpow, colh0, temph0, clumping=1.0):
    x = C.c_double(0.0)
    xa = C.c_double(0.0)
    lib().oracle_doric(C.c_double(xh_old), C.c_double(dt), C.c_double(temp), C.c_double(rhe),
                       C.c_double(phi), C.c_double(bh00), C.c_double(albpow), C.c_double(colh0),
                       C.c_double(temph0), C.c_double(clumping), C.byref(x), C.byref(xa))
    return x.value, xa.value


def do_chemistry(dt, ndens_p, temp, xh_p, xh_av_p, phi, bh00, albpow, colh0, temph0, abu_c):
    xa = C.c_double(xh_av_p)
    xi = C.c_double(0.0)
    nit = lib().oracle_do_chemistry(C.c_double(dt), C.c_double(ndens_p), C.c_double(temp),
                                    C.c_double(xh_p), C.byref(xa), C.byref(xi), C.c_double(phi),
                                    C.c_double(bh00), C.c_double(albpow), C.c_double(colh0),
                                    C.c_double(temph0), C.c_double(abu_c))
    return xi.value, xa.value, nit


def cinterp(pos, srcpos, coldens, sig, flags=0):
    """pos, srcpos: 1-based (3,) ints; coldens (m1,m2,m3) logical.  Returns (cdensi, path)."""
    cd = np.asfortranarray(coldens, dtype=np.float64)
    p = (C.c_int * 3)(*[int(v) for v in pos])
    s = (C.c_int * 3)(*[int(v) for v in srcpos])
    a = C.c_double(0.0)
    b = C.c_double(0.0)
    lib().oracle_cinterp_probe(p, s, _d(cd), C.c_int(cd.shape[0]), C.c_int(cd.shape[1]),
                               C.c_int(cd.shape[2]), C.c_double(sig), C.c_int(flags),
                               C.byref(a), C.byref(b))
    return a.value, b.value


def photoion_rates(normflux, cd_in, cd_out, vfact, sig, thin, thick, minlogtau, dlogtau,
                   heat_thin=None, heat_thick=None, NumTau=None, flags=0):
    thin = np.ascontiguousarray(thin, dtype=np.float64)
    thick = np.ascontiguousarray(thick, dtype=np.float64)
    ht = None if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
    hk = None if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
    if NumTau is None:
        NumTau = thin.shape[0]
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    lib().oracle_photo_rates_probe(C.c_double(normflux), C.c_double(cd_in), C.c_double(cd_out),
                                   C.c_double(vfact), C.c_double(sig), _d(thin), _d(thick),
                                   _opt(ht), _opt(hk), C.c_double(minlogtau), C.c_double(dlogtau),
                                   C.c_int(NumTau), C.c_int(thin.shape[0]), C.c_int(flags),
                                   C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
