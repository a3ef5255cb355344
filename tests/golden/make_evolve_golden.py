"""Golden vectors for the Python orchestration, produced by the REFERENCE'S OWN Python code.

Run in the build container (where /root/reference exists):

    make -C oracle
    python tests/golden/make_evolve_golden.py

What runs: /root/reference/pyc2ray/evolve.py (evolve3D) and raytracing.py (do_raytracing), loaded where they lie
with importlib under a synthetic package (`import pyc2ray` itself fails on astropy, an ordinary ModuleNotFoundError;
these two modules need only numpy and four siblings).  The siblings are the reference's own utils/logutils.py,
utils/sourceutils.py and asora_core.py, loaded the same way; `load_extensions` is the one synthetic module: its
`load_c2ray()` returns ctypes shims over oracle/_ref/libc2ray_ref.so -- the reference's Fortran compiled by
oracle/Makefile -- that behave as the f2py module does (intent(inout) arrays must be Fortran-contiguous float64 and
are modified in place, intent(in) arrays are converted), and its `load_asora()` returns a stand-in for the CUDA
module backed by this repository's C restatement of the ASORA kernel (oracle/liboracle.so; the CUDA sources cannot
be built here), so that the reference's `use_gpu=True` bookkeeping (ravel/reshape, table length as NumTau, ...)
is exercised as well.  Nothing of the reference's text is stored: tests/golden/evolve.npz holds outputs only
(xh_new, phi_ion, iteration counts, the per-iteration convergence numbers of the log); inputs are regenerated
from the seeds in tests/cases.py.
"""
import ctypes as C
import importlib.util
import os
import re
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))

import cases  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import ref_fortran as F  # noqa: E402

REF = "/root/reference/pyc2ray"
PKG = "refpyc2ray"
_dp = C.POINTER(C.c_double)


# ---- f2py-like shims over the compiled reference Fortran ---------------------------------------------------------
def _inout(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.f_contiguous):
        raise ValueError(f"failed in converting argument `{name}' to C/Fortran array: intent(inout) array must be "
                         "contiguous and with a proper type")
    return a


class _RefRaytracing:
    def do_all_sources(self, normflux, srcpos, max_subbox, subboxsize, coldensh_out, sig, dr, ndens, xh_av, phi_ion,
                       phi_heat, loss_fraction, thin, thick, hthin, hthick, minlogtau, dlogtau, r_max_lls):
        N = coldensh_out.shape[0]
        cd, phi, heat, xh = (_inout(coldensh_out, "coldensh_out"), _inout(phi_ion, "phi_ion"),
                             _inout(phi_heat, "phi_heat"), _inout(xh_av, "xh_av"))
        nd = np.asfortranarray(ndens, dtype=np.float64)
        flux = np.ascontiguousarray(normflux, dtype=np.float64)
        pos = np.asfortranarray(np.asarray(srcpos).astype(np.int32))
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in (thin, thick, hthin, hthick)]
        nbox, loss = C.c_int(0), C.c_double(0.0)
        d = lambda a: a.ctypes.data_as(_dp)
        r = lambda x: C.byref(C.c_double(x))
        i = lambda x: C.byref(C.c_int(x))
        F.lib()._QMraytracingPdo_all_sources(
            d(flux), pos.ctypes.data_as(C.POINTER(C.c_int32)), i(max_subbox), i(subboxsize), d(cd), r(sig), r(dr), d(nd),
            d(xh), d(phi), d(heat), C.byref(nbox), C.byref(loss), C.byref(C.c_float(loss_fraction)), d(tabs[0]), d(tabs[1]),
            d(tabs[2]), d(tabs[3]), r(minlogtau), r(dlogtau), r(r_max_lls), i(tabs[0].shape[0]), i(flux.shape[0]), i(N), i(N), i(N))
        return nbox.value, loss.value


class _RefChemistry:
    def global_pass(self, dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c):
        xa, xi = _inout(xh_av, "xh_av"), _inout(xh_intermed, "xh_intermed")
        shp = xa.shape
        f = lambda a: np.asfortranarray(a, dtype=np.float64)
        nd, tp, x0, ph = f(ndens), f(temp), f(xh), f(phi_ion)
        conv = C.c_int(0)
        d = lambda a: a.ctypes.data_as(_dp)
        r = lambda x: C.byref(C.c_double(x))
        i = lambda x: C.byref(C.c_int(x))
        F.lib()._QMchemistryPglobal_pass(r(dt), d(nd), d(tp), d(x0), d(xa), d(xi), d(ph), r(bh00), r(albpow), r(colh0),
                                         r(temph0), r(abu_c), C.byref(conv), i(shp[0]), i(shp[1]), i(shp[2]))
        return conv.value


class _RefC2Ray:
    raytracing = _RefRaytracing()
    chemistry = _RefChemistry()


class _OracleAsora:
    """Stand-in for the CUDA extension module (src/asora/python_module.cu:153-161), backed by oracle/liboracle.so."""

    def device_init(self, N, batch):
        self.N = N

    def device_close(self):
        pass

    def photo_table_to_device(self, thin, thick, NumTau):
        self.thin, self.thick = np.array(thin[:NumTau]), np.array(thick[:NumTau])

    def source_data_to_device(self, pos, flux, n):
        self.pos, self.flux = np.array(pos[:3 * n]), np.array(flux[:n])

    def density_to_device(self, ndens_flat, N):
        self.ndens = np.array(ndens_flat).reshape(N, N, N)

    def do_all_sources(self, R, cd_flat, sig, dr, ndens_flat, xh_flat, phi_flat, NumSrc, N, minlogtau, dlogtau, NumTau):
        r = O.asora_do_all_sources(R, sig, dr, self.ndens, np.asarray(xh_flat).reshape(N, N, N), self.pos[:3 * NumSrc],
                                   self.flux[:NumSrc], self.thin, self.thick, minlogtau, dlogtau, NumTau=NumTau,
                                   flags=O.ASORA_MODE)
        phi_flat[:] = r["phi_ion"].ravel()


def load_reference_modules():
    """The reference's evolve / raytracing / asora_core modules, loaded from their files under a synthetic package."""
    pkg = types.ModuleType(PKG)
    pkg.__path__ = []
    sys.modules[PKG] = pkg
    ext = types.ModuleType(PKG + ".load_extensions")
    asora, c2ray = _OracleAsora(), _RefC2Ray()
    ext.load_c2ray = lambda: c2ray
    ext.load_asora = lambda: asora
    sys.modules[PKG + ".load_extensions"] = ext

    def from_file(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    utils = types.ModuleType(PKG + ".utils")
    utils.__path__ = []
    sys.modules[PKG + ".utils"] = utils
    logutils = from_file(PKG + ".utils.logutils", os.path.join(REF, "utils", "logutils.py"))
    sourceutils = from_file(PKG + ".utils.sourceutils", os.path.join(REF, "utils", "sourceutils.py"))
    utils.printlog = logutils.printlog
    utils.format_sources = sourceutils.format_sources
    core = from_file(PKG + ".asora_core", os.path.join(REF, "asora_core.py"))
    ev = from_file(PKG + ".evolve", os.path.join(REF, "evolve.py"))
    rt = from_file(PKG + ".raytracing", os.path.join(REF, "raytracing.py"))
    return ev, rt, core


def convergence_rows(log_text):
    """[(conv_flag, relative change as printed)] per outer iteration, from the reference's own log lines."""
    rows = re.findall(r"Number of non-converged points: (\d+) of \d+ .*Relative change in ionfrac:\s*([0-9.eE+-]+)", log_text)
    return np.array([(int(a), float(b)) for a, b in rows], dtype=np.float64).reshape(-1, 2)


def main():
    assert F.available(), "build oracle/_ref first: make -C oracle"
    ev, rt, core = load_reference_modules()
    out = {}
    for name in cases.EVOLVE_CASES:
        c = cases.evolve_case(name)
        N = c["N"]
        if c["use_gpu"]:
            core.device_init(N, 8)
            core.photo_table_to_device(c["thin"], c["thick"])
        xh = c["xh"]
        nsteps = c["steps"]
        with tempfile.TemporaryDirectory() as tmp:
            log = os.path.join(tmp, "log")
            for step in range(nsteps):
                xh_new, phi = ev.evolve3D(c["dt"], c["dr"], c["flux"], c["pos"], c["use_gpu"], c["max_subbox"],
                                          c["subboxsize"], c["loss_fraction"], c["temp"], c["ndens"], xh, c["thin"],
                                          c["thick"], cases.MINLOGTAU, c["dlogtau"], c["R"], c["convergence_fraction"],
                                          cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C,
                                          logfile=log, quiet=True)
                rows = convergence_rows(open(log).read())
                done = sum(len(out[k]) for k in out if k.startswith(f"{name}__rows"))
                out[f"{name}__rows{step}"] = rows[done:]
                out[f"{name}__xh{step}"] = np.ascontiguousarray(xh_new)        # logical [i,j,k]
                out[f"{name}__phi{step}"] = np.ascontiguousarray(phi)
                out[f"{name}__orders{step}"] = np.array([xh_new.flags.f_contiguous and not xh_new.flags.c_contiguous,
                                                         phi.flags.f_contiguous and not phi.flags.c_contiguous])
                xh = xh_new
        print(f"{name}: iterations per step {[len(out[f'{name}__rows{s}']) for s in range(nsteps)]}, "
              f"<x> = {out[f'{name}__xh{nsteps - 1}'].mean():.6f}")
    # the standalone raytracing call (CPU branch; the reference's GPU branch returns an undefined name, raytracing.py:108)
    for name in cases.RAYTRACING_CASES:
        c = cases.evolve_case(name)
        with tempfile.TemporaryDirectory() as tmp:
            phi, nbox, loss = rt.do_raytracing(c["dr"], c["flux"], c["pos"], False, c["max_subbox"], c["subboxsize"],
                                               c["loss_fraction"], c["ndens"], np.asfortranarray(c["xh"]), c["thin"], c["thick"],
                                               c["heat_thin"], c["heat_thick"], cases.MINLOGTAU, c["dlogtau"], c["R"], cases.SIG,
                                               logfile=os.path.join(tmp, "log"), quiet=True, stats=True)
            phi2, heat = rt.do_raytracing(c["dr"], c["flux"], c["pos"], False, c["max_subbox"], c["subboxsize"],
                                          c["loss_fraction"], c["ndens"], np.asfortranarray(c["xh"]), c["thin"], c["thick"],
                                          c["heat_thin"], c["heat_thick"], cases.MINLOGTAU, c["dlogtau"], c["R"], cases.SIG,
                                          logfile=os.path.join(tmp, "log"), quiet=True)
        assert np.array_equal(phi, phi2)
        out[f"rt_{name}__phi"] = np.ascontiguousarray(phi)
        out[f"rt_{name}__heat"] = np.ascontiguousarray(heat)
        out[f"rt_{name}__stats"] = np.array([nbox, loss])
        print(f"do_raytracing {name}: nsubbox {nbox}, photon loss {loss:.3e}")
    np.savez_compressed(os.path.join(HERE, "evolve.npz"), **out)
    print("written", os.path.join(HERE, "evolve.npz"))


if __name__ == "__main__":
    main()
