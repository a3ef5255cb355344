"""Golden radiation tables, produced by the REFERENCE'S OWN Python code.

Run in the build container (where /root/reference exists):

    python tests/golden/make_radiation_golden.py

What runs: /root/reference/pyc2ray/radiation/common.py (make_tau_table) and blackbody.py (BlackBodySource), loaded
where they lie with importlib.  Both import `astropy.constants` at module level, and astropy is not installed here
(an ordinary ModuleNotFoundError).  blackbody.py uses three of its constants -- h, Ryd, c -- ONLY in the heating
integrands (ref: blackbody.py:15-16,60-66: `hplanck`, `ion_freq_HI`); the photo-ionisation tables this script stores
(make_photo_table, ref: blackbody.py:71-77) and make_tau_table (common.py:13-37) do not touch them.  So the import is
satisfied by a namespace holding those three CODATA-2018 numbers, and NO heating table is generated or stored: the
fixture pins exactly the part of the reference that runs unmodified.

tests/golden/radiation.npz: the tau table for (-20, 4, 2000) and (thin, thick) for Teff in {5e3, 5e4, 1e5} K, grey and
power-law cross sections, integration limits and normalisation of ref: pyc2ray/c2ray_base.py:400-417 (13.598 eV to
10 x 54.416 eV, S_star = 1e48).  Outputs only.
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pyc2ray/radiation"
EV2FR = 0.241838e15
TEFFS = (5e3, 5e4, 1e5)


def _astropy_constants_namespace():
    """h, Ryd, c with the `.cgs.value` / arithmetic the reference's module-level lines need (blackbody.py:3-5,15-16)."""
    class _Q:
        def __init__(self, v):
            self.value = v
            self.cgs = self

        def __mul__(self, other):
            return _Q(self.value * (other.value if isinstance(other, _Q) else other))
    consts = types.ModuleType("astropy.constants")
    consts.h = _Q(6.62607015e-27)            # erg s
    consts.Ryd = _Q(109737.31568160)         # 1/cm
    consts.c = _Q(2.99792458e10)             # cm/s
    consts.k_B = _Q(1.380649e-16)
    astropy = types.ModuleType("astropy")
    astropy.constants = consts
    return astropy, consts


def load_reference_radiation():
    had = {k: sys.modules.get(k) for k in ("astropy", "astropy.constants")}
    if had["astropy"] is None:
        sys.modules["astropy"], sys.modules["astropy.constants"] = _astropy_constants_namespace()
    try:
        mods = []
        for name in ("common", "blackbody"):
            spec = importlib.util.spec_from_file_location("refradiation_" + name, os.path.join(REF, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mods.append(mod)
    finally:
        for k, v in had.items():
            if v is None:
                sys.modules.pop(k, None)
    return mods


def main():
    common, blackbody = load_reference_radiation()
    tau, dlog = common.make_tau_table(-20.0, 4.0, 2000)
    out = {"tau": tau, "dlogtau": np.array(dlog), "teffs": np.array(TEFFS)}
    f1, f2 = EV2FR * 13.598, 10 * EV2FR * 54.416
    for teff in TEFFS:
        for grey in (True, False):
            src = blackbody.BlackBodySource(teff, grey, EV2FR * 13.598, 2.8)
            thin, thick = src.make_photo_table(tau, f1, f2, 1e48)
            key = f"T{teff:g}_{'grey' if grey else 'pl'}"
            out[key + "_thin"], out[key + "_thick"] = thin, thick
            out[key + "_Rstar"] = np.array(src.R_star)
            print(f"{key}: thin[0] = {thin[0]:.6e}, thick[0] = {thick[0]:.6e}, thick[1200] = {thick[1200]:.6e}")
    np.savez_compressed(os.path.join(HERE, "radiation.npz"), **out)
    print("written", os.path.join(HERE, "radiation.npz"))


if __name__ == "__main__":
    main()
