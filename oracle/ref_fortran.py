"""ctypes binding of oracle/_ref/libc2ray_ref.so (TEST INFRASTRUCTURE, see oracle/__init__.py).

The shared object is the reference's own src/c2ray/{photorates,raytracing,chemistry}.f90
compiled in place by oracle/Makefile with flang.  Every Fortran dummy argument is passed by
reference in declaration order (raytracing.f90:52-56, chemistry.f90:13, ...); arrays are
explicit-shape, i.e. bare pointers to Fortran-ordered float64 / int32 storage.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libc2ray_ref.so")

_dp = C.POINTER(C.c_double)


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libc2ray_ref.so not built (run `make -C oracle` "
                               "where /root/reference is present)")
        _lib = C.CDLL(_PATH)
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _r(x):
    return C.byref(C.c_double(x))


def _i(x):
    return C.byref(C.c_int(x))


def do_all_sources(normflux, srcpos, max_subbox, subboxsize, sig, dr, ndens, xh_av, loss_fraction,
                   thin, thick, minlogtau, dlogtau, R_max_LLS, heat_thin=None, heat_thick=None,
                   NumTau=None, copy_xh_av=True):
    """raytracing.f90:52 do_all_sources.  Same conventions as oracle.oracle.do_all_sources.
    copy_xh_av=False hands the caller's Fortran-ordered xh_av over as it is (the dummy is intent(inout), :68, but the
    routine never writes it): bench.py's CPU workers share one read-only mapping of the grid that way."""
    N = ndens.shape[0]
    nd = np.asfortranarray(ndens, dtype=np.float64)
    xh = np.asfortranarray(xh_av, dtype=np.float64)
    if copy_xh_av or not xh.flags.f_contiguous:
        xh = xh.copy(order="F")
    flux = np.ascontiguousarray(normflux, dtype=np.float64)
    pos = np.asfortranarray(np.asarray(srcpos).astype(np.int32))
    thin = np.ascontiguousarray(thin, dtype=np.float64)
    thick = np.ascontiguousarray(thick, dtype=np.float64)
    if NumTau is None:
        NumTau = thin.shape[0]
    ht = np.zeros(thin.shape[0]) if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
    hk = np.zeros(thin.shape[0]) if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
    phi = np.zeros((N, N, N), order="F")
    heat = np.zeros((N, N, N), order="F")
    cd = np.zeros((N, N, N), order="F")
    nbox = C.c_int(0)
    loss = C.c_double(0.0)
    lib()._QMraytracingPdo_all_sources(
        _d(flux), pos.ctypes.data_as(C.POINTER(C.c_int32)), _i(max_subbox), _i(subboxsize), _d(cd),
        _r(sig), _r(dr), _d(nd), _d(xh), _d(phi), _d(heat), C.byref(nbox), C.byref(loss),
        C.byref(C.c_float(loss_fraction)), _d(thin), _d(thick), _d(ht), _d(hk),
        _r(minlogtau), _r(dlogtau), _r(R_max_LLS), _i(NumTau), _i(flux.shape[0]), _i(N), _i(N), _i(N))
    return dict(phi_ion=phi, phi_heat=heat, coldens=cd, nsubbox=nbox.value, photon_loss=loss.value)


def global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c):
    """chemistry.f90:13 global_pass.  Returns (xh_av_new, xh_intermed_new, conv_flag) as NEW
    arrays shaped like the inputs (logical [i,j,k])."""
    shp = ndens.shape
    f = lambda a: np.asfortranarray(a, dtype=np.float64).copy(order="F")
    nd, tp, x0, xa, xi, ph = map(f, (ndens, temp, xh, xh_av, xh_intermed, phi_ion))
    conv = C.c_int(0)
    lib()._QMchemistryPglobal_pass(_r(dt), _d(nd), _d(tp), _d(x0), _d(xa), _d(xi), _d(ph),
                                   _r(bh00), _r(albpow), _r(colh0), _r(temph0), _r(abu_c),
                                   C.byref(conv), _i(shp[0]), _i(shp[1]), _i(shp[2]))
    return xa, xi, conv.value


def doric(xh_old, dt, temp, rhe, phi, bh00, albpow, colh0, temph0, clumping=1.0):
    x, xa = C.c_double(0.0), C.c_double(0.0)
    lib()._QMchemistryPdoric(_r(xh_old), _r(dt), _r(temp), _r(rhe), _r(phi), _r(bh00), _r(albpow),
                             _r(colh0), _r(temph0), _r(clumping), C.byref(x), C.byref(xa))
    return x.value, xa.value


def do_chemistry(dt, ndens_p, temp, xh_p, xh_av_p, phi, bh00, albpow, colh0, temph0, abu_c):
    xp = C.c_double(xh_p)
    xa = C.c_double(xh_av_p)
    xi = C.c_double(0.0)
    lib()._QMchemistryPdo_chemistry(_r(dt), _r(ndens_p), _r(temp), C.byref(xp), C.byref(xa),
                                    C.byref(xi), _r(phi), _r(bh00), _r(albpow), _r(colh0),
                                    _r(temph0), _r(abu_c))
    return xi.value, xa.value


def cinterp(pos, srcpos, coldens, sig):
    cd = np.asfortranarray(coldens, dtype=np.float64)
    p = (C.c_int * 3)(*[int(v) for v in pos])
    s = (C.c_int * 3)(*[int(v) for v in srcpos])
    a, b = C.c_double(0.0), C.c_double(0.0)
    lib()._QMraytracingPcinterp(p, s, C.byref(a), C.byref(b), _d(cd), _r(sig),
                                _i(cd.shape[0]), _i(cd.shape[1]), _i(cd.shape[2]))
    return a.value, b.value


def photoion_rates(normflux, cd_in, cd_out, vfact, sig, thin, thick, minlogtau, dlogtau,
                   heat_thin=None, heat_thick=None, NumTau=None):
    thin = np.ascontiguousarray(thin, dtype=np.float64)
    thick = np.ascontiguousarray(thick, dtype=np.float64)
    if NumTau is None:
        NumTau = thin.shape[0]
    ht = np.zeros(thin.shape[0]) if heat_thin is None else np.ascontiguousarray(heat_thin, dtype=np.float64)
    hk = np.zeros(thin.shape[0]) if heat_thick is None else np.ascontiguousarray(heat_thick, dtype=np.float64)
    a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
    lib()._QMphotoratesPphotoion_rates(_r(normflux), _r(cd_in), _r(cd_out), _r(vfact), _r(sig),
                                       C.byref(a), C.byref(b), C.byref(c), _d(thin), _d(thick),
                                       _d(ht), _d(hk), _r(minlogtau), _r(dlogtau), _i(NumTau))
    return a.value, b.value, c.value


def photoion_rates_test(normflux, cd_in, cd_out, vfact, nHI, sig):
    a, b = C.c_double(0), C.c_double(0)
    lib()._QMphotoratesPphotoion_rates_test(_r(normflux), _r(cd_in), _r(cd_out), _r(vfact), _r(nHI),
                                            _r(sig), C.byref(a), C.byref(b))
    return a.value, b.value
