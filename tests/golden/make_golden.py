"""Generate the committed golden vectors from the REFERENCE ITSELF.

Run in the build container (where /root/reference exists):

    make -C oracle          # compiles the reference Fortran in place -> oracle/_ref/
    python tests/golden/make_golden.py

Every expected output below is produced by oracle/_ref/libc2ray_ref.so, i.e. by the reference's
own src/c2ray/*.f90 compiled with flang -- NOT by this repository's restatement.  Inputs are the
seeded cases of tests/cases.py (regenerated at test time); only outputs (and the few explicit
probe inputs) are stored.  The fixtures are data: no reference source text is kept.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import cases  # noqa: E402
from oracle import ref_fortran as F  # noqa: E402


def gen_cinterp():
    """cinterp on every cell of a 9^3 box around a central source (all 26 neighbour classes,
    ties, axes, planes) plus an off-centre source with periodic wrap."""
    rng = np.random.default_rng(42)
    out = {}
    for tag, M, src in (("c9", 9, (5, 5, 5)), ("w7", 7, (2, 6, 1))):
        cd = 10 ** rng.uniform(15, 19, size=(M, M, M))
        res = []
        lo, hi = -(M // 2), M // 2 - 1 + M % 2
        for di in range(lo, hi + 1):
            for dj in range(lo, hi + 1):
                for dk in range(lo, hi + 1):
                    if di == dj == dk == 0:
                        continue
                    pos = (src[0] + di, src[1] + dj, src[2] + dk)
                    c, p = F.cinterp(pos, src, cd, cases.SIG)
                    res.append((di, dj, dk, c, p))
        out[tag + "_cd"] = cd
        out[tag + "_src"] = np.array(src)
        out[tag + "_res"] = np.array(res)
    np.savez_compressed(os.path.join(HERE, "cinterp.npz"), **out)


def gen_rates():
    thin, thick, dlog = cases.soft_tables(2000)
    hthin, hthick = 1e-11 * thin[::-1].copy(), 1e-11 * thick * 0.5
    NT = 2000                                     # NumTau = len-1, as the benchmark passes it
    taus = [0.0, 1e-25, 1e-20, 3e-13, 1e-7, 0.999e-7, 1e-3, 1.0, 7.7, 1e2, 9.9e3]
    rows = []
    for tin in taus:
        for dt_ in [0.0, 1e-12, 5e-8, 1.00000001e-7, 1e-3, 0.5, 40.0]:
            cin, cout = tin / cases.SIG, (tin + dt_) / cases.SIG
            a, b, c = F.photoion_rates(2.5, cin, cout, 3.3e70, cases.SIG, thin, thick,
                                       cases.MINLOGTAU, dlog, hthin, hthick, NumTau=NT)
            g = F.photoion_rates_test(2.5, cin, cout, 3.3e70, 1e-3, cases.SIG)
            rows.append((cin, cout, a, b, c, g[0], g[1]))
    np.savez_compressed(os.path.join(HERE, "rates.npz"), rows=np.array(rows), NumTau=NT,
                        normflux=2.5, vfact=3.3e70)


def gen_chem_points():
    rows_d, rows_c = [], []
    for x0 in [1e-14, 2e-4, 0.1, 0.9, 1.0 - 1e-9]:
        for phi in [0.0, 1e-18, 1e-13, 1e-9, 1e-2]:
            for n in [1e-6, 1e-3, 1.0]:
                for dt in [1e3, 3.15576e13, 1e17]:
                    for T in [1e3, 1e4, 5e4]:
                        rhe = n * (x0 + cases.ABU_C)
                        a, b = F.doric(x0, dt, T, rhe, phi, cases.BH00, cases.ALBPOW, cases.COLH0,
                                       cases.TEMPH0)
                        rows_d.append((x0, dt, T, rhe, phi, a, b))
                        xi, xa = F.do_chemistry(dt, n, T, x0, x0, phi, cases.BH00, cases.ALBPOW,
                                                cases.COLH0, cases.TEMPH0, cases.ABU_C)
                        rows_c.append((dt, n, T, x0, phi, xi, xa))
    np.savez_compressed(os.path.join(HERE, "chem_points.npz"), doric=np.array(rows_d),
                        do_chemistry=np.array(rows_c))


def gen_raytrace():
    out = {}
    for name in cases.RT_CASES:
        for tables in ("grey", "soft"):
            c = cases.rt_case(name, tables)
            N = c["N"]
            r = F.do_all_sources(c["flux"], c["pos"], max_subbox=1000, subboxsize=N, sig=c["sig"],
                                 dr=c["dr"], ndens=c["ndens"], xh_av=c["xh"], loss_fraction=0.0,
                                 thin=c["thin"], thick=c["thick"], minlogtau=c["minlogtau"],
                                 dlogtau=c["dlogtau"], R_max_LLS=c["R"],
                                 NumTau=c["thin"].shape[0] - 1)
            key = f"{name}__{tables}"
            out[key + "__phi"] = np.ascontiguousarray(r["phi_ion"])      # logical [i,j,k]
            out[key + "__cd"] = np.ascontiguousarray(r["coldens"])       # last source
            out[key + "__stats"] = np.array([r["nsubbox"], r["photon_loss"]])
    # sub-box growth with early stop (loss_fraction > 0) on one case
    c = cases.rt_case("l32_5src_R10", "grey")
    r = F.do_all_sources(c["flux"], c["pos"], max_subbox=12, subboxsize=3, sig=c["sig"], dr=c["dr"],
                         ndens=c["ndens"], xh_av=c["xh"], loss_fraction=1e-2, thin=c["thin"],
                         thick=c["thick"], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"],
                         R_max_LLS=1000.0, NumTau=c["thin"].shape[0] - 1)
    out["subbox__phi"] = np.ascontiguousarray(r["phi_ion"])
    out["subbox__stats"] = np.array([r["nsubbox"], r["photon_loss"]])
    np.savez_compressed(os.path.join(HERE, "raytrace.npz"), **out)


def gen_subbox():
    """The reference's CPU raytracer with sub-box growth, heating tables and unequal fluxes."""
    out = {}
    for name in cases.SUBBOX_CASES:
        c = cases.subbox_case(name)
        r = F.do_all_sources(c["flux"], c["pos"], max_subbox=c["max_subbox"], subboxsize=c["subboxsize"],
                             sig=c["sig"], dr=c["dr"], ndens=c["ndens"], xh_av=c["xh"],
                             loss_fraction=c["loss_fraction"], thin=c["thin"], thick=c["thick"],
                             minlogtau=c["minlogtau"], dlogtau=c["dlogtau"], R_max_LLS=c["R"],
                             heat_thin=c["heat_thin"], heat_thick=c["heat_thick"], NumTau=c["thin"].shape[0] - 1)
        out[name + "__phi"] = np.ascontiguousarray(r["phi_ion"])
        out[name + "__heat"] = np.ascontiguousarray(r["phi_heat"])
        out[name + "__cd"] = np.ascontiguousarray(r["coldens"])
        out[name + "__stats"] = np.array([r["nsubbox"], r["photon_loss"]])
    np.savez_compressed(os.path.join(HERE, "subbox.npz"), **out)


def gen_global_pass():
    out = {}
    for N, seed in ((16, 21), (12, 22)):
        c = cases.chem_case(N, seed)
        xa, xi, conv = F.global_pass(c["dt"], c["ndens"], c["temp"], c["xh"], c["xh_av"],
                                     c["xh_intermed"], c["phi_ion"], c["bh00"], c["albpow"],
                                     c["colh0"], c["temph0"], c["abu_c"])
        out[f"n{N}_xh_av"] = np.ascontiguousarray(xa)
        out[f"n{N}_xh_intermed"] = np.ascontiguousarray(xi)
        out[f"n{N}_conv"] = np.array(conv)
    np.savez_compressed(os.path.join(HERE, "global_pass.npz"), **out)


if __name__ == "__main__":
    assert F.available(), "build oracle/_ref first: make -C oracle"
    gen_cinterp()
    gen_rates()
    gen_chem_points()
    gen_raytrace()
    gen_subbox()
    gen_global_pass()
    print("golden vectors written to", HERE)
