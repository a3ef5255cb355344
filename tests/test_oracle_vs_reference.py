"""CPU, only where oracle/_ref exists (the build container): randomised comparison of the C
restatement with the reference's own compiled Fortran, beyond the committed golden vectors."""
import numpy as np
import pytest

import cases
from oracle import oracle as O
from oracle import ref_fortran as F

pytestmark = pytest.mark.skipif(not F.available(), reason="oracle/_ref/libc2ray_ref.so not built here")


@pytest.mark.parametrize("seed", range(6))
def test_random_raytrace_configurations(seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(8, 22))
    ns = int(rng.integers(1, 5))
    nd, xh, dr = cases.grid(N, "lognormal", 2000 + seed, float(10 ** rng.uniform(-3, 1)))
    pos, flux = cases.sources(N, ns, 3000 + seed, flux=float(rng.uniform(0.5, 50)))
    thin, thick, dlog = cases.soft_tables()
    kw = dict(max_subbox=int(rng.integers(2, 40)), subboxsize=int(rng.integers(1, 9)), sig=cases.SIG, dr=dr, ndens=nd,
              xh_av=xh, loss_fraction=float(rng.choice([0.0, 1e-3, 1e-1])), thin=thin, thick=thick,
              minlogtau=cases.MINLOGTAU, dlogtau=dlog, R_max_LLS=1000.0, NumTau=thin.shape[0] - 1)
    a, b = O.do_all_sources(flux, pos, **kw), F.do_all_sources(flux, pos, **kw)
    assert np.array_equal(a["phi_ion"], b["phi_ion"])
    assert np.array_equal(a["coldens"], b["coldens"])
    assert (a["nsubbox"], a["photon_loss"]) == (b["nsubbox"], b["photon_loss"])


@pytest.mark.parametrize("seed", range(4))
def test_random_chemistry_passes(seed):
    c = cases.chem_case(10 + seed, 4000 + seed, dt=float(10 ** np.random.default_rng(seed).uniform(10, 16)))
    args = (c["dt"], c["ndens"], c["temp"], c["xh"], c["xh_av"], c["xh_intermed"], c["phi_ion"], c["bh00"],
            c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    xa, xi, conv, _ = O.global_pass(*args)
    xa2, xi2, conv2 = F.global_pass(*args)
    assert np.array_equal(xa, xa2) and np.array_equal(xi, xi2) and conv == conv2


@pytest.mark.parametrize("seed", range(3))
def test_heating_grid_of_the_fortran_path(seed):
    N = 12 + seed
    nd, xh, dr = cases.grid(N, "lognormal", 5000 + seed, 0.07)
    pos, flux = cases.sources(N, 3, 5100 + seed, flux=2.0)
    thin, thick, dlog = cases.soft_tables()
    ht, hk = 3e-11 * thin[::-1].copy(), 2e-11 * thick
    kw = dict(max_subbox=1000, subboxsize=N, sig=cases.SIG, dr=dr, ndens=nd, xh_av=xh, loss_fraction=0.0, thin=thin,
              thick=thick, minlogtau=cases.MINLOGTAU, dlogtau=dlog, R_max_LLS=1000.0, heat_thin=ht, heat_thick=hk,
              NumTau=thin.shape[0] - 1)
    a, b = O.do_all_sources(flux, pos, **kw), F.do_all_sources(flux, pos, **kw)
    assert np.array_equal(a["phi_heat"], b["phi_heat"]) and np.any(a["phi_heat"])
    assert np.array_equal(a["phi_ion"], b["phi_ion"])


def test_heating_tables_and_grey_rates():
    thin, thick, dlog = cases.soft_tables()
    ht, hk = 3e-11 * thin, 2e-11 * thick
    for cin, cout in [(0.0, 1e15), (1e17, 1.00000001e17), (3e18, 9e18), (1e21, 1.2e21)]:
        a = O.photoion_rates(4.0, cin, cout, 1e70, cases.SIG, thin, thick, cases.MINLOGTAU, dlog, ht, hk,
                             NumTau=thin.shape[0] - 1)
        b = F.photoion_rates(4.0, cin, cout, 1e70, cases.SIG, thin, thick, cases.MINLOGTAU, dlog, ht, hk,
                             NumTau=thin.shape[0] - 1)
        assert a == b
        g = O.photoion_rates(4.0, cin, cout, 1e70, cases.SIG, thin, thick, cases.MINLOGTAU, dlog, flags=O.GREY)
        assert g[:2] == F.photoion_rates_test(4.0, cin, cout, 1e70, 1e-3, cases.SIG)
