"""CPU: the C restatement (oracle/) against the committed golden vectors, which were produced by
the reference's own Fortran (tests/golden/make_golden.py).  Bit-exact unless stated."""
import os

import numpy as np
import pytest

import cases
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_cinterp_all_neighbour_classes():
    g = _load("cinterp.npz")
    for tag in ("c9", "w7"):
        cd, src, res = g[tag + "_cd"], g[tag + "_src"], g[tag + "_res"]
        for di, dj, dk, c_ref, p_ref in res:
            pos = (src[0] + int(di), src[1] + int(dj), src[2] + int(dk))
            c, p = O.cinterp(pos, src, cd, cases.SIG)
            assert c == c_ref and p == p_ref, (tag, di, dj, dk)


def test_photo_rates_lattice():
    g = _load("rates.npz")
    thin, thick, dlog = cases.soft_tables(2000)
    hthin, hthick = 1e-11 * thin[::-1].copy(), 1e-11 * thick * 0.5
    for cin, cout, a, b, c, ga, gb in g["rows"]:
        r = O.photoion_rates(float(g["normflux"]), cin, cout, float(g["vfact"]), cases.SIG, thin, thick,
                             cases.MINLOGTAU, dlog, hthin, hthick, NumTau=int(g["NumTau"]))
        assert r == (a, b, c)
        rg = O.photoion_rates(float(g["normflux"]), cin, cout, float(g["vfact"]), cases.SIG, thin, thick,
                              cases.MINLOGTAU, dlog, NumTau=int(g["NumTau"]), flags=O.GREY)
        assert rg[:2] == (ga, gb)


def test_doric_and_do_chemistry_lattice():
    g = _load("chem_points.npz")
    for x0, dt, T, rhe, phi, a, b in g["doric"]:
        assert O.doric(x0, dt, T, rhe, phi, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0) == (a, b)
    for dt, n, T, x0, phi, xi, xa in g["do_chemistry"]:
        r = O.do_chemistry(dt, n, T, x0, x0, phi, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0,
                           cases.ABU_C)
        assert r[:2] == (xi, xa)


@pytest.mark.parametrize("name", list(cases.RT_CASES))
@pytest.mark.parametrize("tables", ["grey", "soft"])
def test_fortran_path_raytrace(name, tables):
    g = _load("raytrace.npz")
    c = cases.rt_case(name, tables)
    r = O.do_all_sources(c["flux"], c["pos"], max_subbox=1000, subboxsize=c["N"], sig=c["sig"], dr=c["dr"],
                         ndens=c["ndens"], xh_av=c["xh"], loss_fraction=0.0, thin=c["thin"],
                         thick=c["thick"], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"],
                         R_max_LLS=c["R"], NumTau=c["thin"].shape[0] - 1)
    key = f"{name}__{tables}"
    assert np.array_equal(r["phi_ion"], g[key + "__phi"])
    assert np.array_equal(r["coldens"], g[key + "__cd"])
    assert r["nsubbox"] == int(g[key + "__stats"][0])
    assert r["photon_loss"] == g[key + "__stats"][1]


def test_fortran_path_subbox_early_stop():
    g = _load("raytrace.npz")
    c = cases.rt_case("l32_5src_R10", "grey")
    r = O.do_all_sources(c["flux"], c["pos"], max_subbox=12, subboxsize=3, sig=c["sig"], dr=c["dr"],
                         ndens=c["ndens"], xh_av=c["xh"], loss_fraction=1e-2, thin=c["thin"],
                         thick=c["thick"], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"],
                         R_max_LLS=1000.0, NumTau=c["thin"].shape[0] - 1)
    assert np.array_equal(r["phi_ion"], g["subbox__phi"])
    assert r["nsubbox"] == int(g["subbox__stats"][0])
    assert r["photon_loss"] == g["subbox__stats"][1]


@pytest.mark.parametrize("name", list(cases.SUBBOX_CASES))
def test_fortran_path_subbox_growth_heating_unequal_fluxes(name):
    """Sub-box growth with early stop, heating tables and unequal fluxes (the reference rates every source with
    the last source's flux, raytracing.f90:500,503): bit-identical to the reference's own output."""
    c = cases.subbox_case(name)
    g = np.load(os.path.join(G, "subbox.npz"))
    r = O.do_all_sources(c["flux"], c["pos"], max_subbox=c["max_subbox"], subboxsize=c["subboxsize"], sig=c["sig"],
                         dr=c["dr"], ndens=c["ndens"], xh_av=c["xh"], loss_fraction=c["loss_fraction"], thin=c["thin"],
                         thick=c["thick"], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"], R_max_LLS=c["R"],
                         heat_thin=c["heat_thin"], heat_thick=c["heat_thick"], NumTau=c["thin"].shape[0] - 1)
    assert np.array_equal(r["phi_ion"], g[name + "__phi"])
    assert np.array_equal(r["phi_heat"], g[name + "__heat"])
    assert np.array_equal(r["coldens"], g[name + "__cd"])
    assert r["nsubbox"] == int(g[name + "__stats"][0])
    assert r["photon_loss"] == g[name + "__stats"][1]


@pytest.mark.parametrize("N,seed", [(16, 21), (12, 22)])
def test_global_pass(N, seed):
    g = _load("global_pass.npz")
    c = cases.chem_case(N, seed)
    xa, xi, conv, its = O.global_pass(c["dt"], c["ndens"], c["temp"], c["xh"], c["xh_av"], c["xh_intermed"],
                                      c["phi_ion"], c["bh00"], c["albpow"], c["colh0"], c["temph0"],
                                      c["abu_c"])
    assert np.array_equal(xa, g[f"n{N}_xh_av"])
    assert np.array_equal(xi, g[f"n{N}_xh_intermed"])
    assert conv == int(g[f"n{N}_conv"])
    assert its >= N ** 3


# --- known-answer test of the reference's chemistry tutorial -----------------------------------
def run_tutorial(global_pass_xint):
    """tutorials/chemistry_solver.ipynb cells 3,5.  hydrogenODE (pyc2ray/chemistry.py:43-95) passes
    the SAME array as xh, xh_av and xh_intermed (aliased intent(inout) dummies: which store
    survives is compiler-dependent).  The value the notebook prints, <x> = 0.127, is reproduced by
    the end-of-step fraction xh_intermed (0.12674); the time-average would give 0.089.
    `global_pass_xint(dt,ndens,temp,xh,xh_av,xh_intermed,phi,...)` must return the xh_intermed grid."""
    mesh = (10, 10, 10)
    np.random.seed(2023)
    ndens = np.random.normal(loc=1e-7, scale=1e-8, size=mesh)
    temp = np.ones(mesh) * 1e4
    xh = np.random.uniform(low=0, high=0.1, size=mesh)
    phi = np.random.uniform(low=1e-13, high=1e-12, size=mesh)
    dt = 50 * 3.15576e7                       # 50 yr (astropy Julian year)
    temph0 = 13.598 * 11604.518121550082      # (13.598 eV / k_B) in K
    series = [xh.mean()]
    for _ in range(100):
        xh = global_pass_xint(dt, ndens, temp, xh, xh, xh, phi, 2.59e-13, -0.7, 1.3e-8, temph0, 7.1e-7)
        series.append(xh.mean())
    return series


def test_chemistry_tutorial_known_answer():
    s = run_tutorial(lambda *a: O.global_pass(*a)[1])
    assert round(s[0], 3) == 0.050           # printed by the notebook, cell 5
    assert round(s[-1], 3) == 0.127          # printed by the notebook, cell 5
