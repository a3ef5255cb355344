"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the product
API, i.e. through the C-ABI of libasora_hip.so; the oracle and the golden vectors are the checkers.

Tolerances (float64 path; north star bar: 1e-5 relative on ionised fraction and column density):
  * column density vs oracle ............ 1e-12  (pure interpolation arithmetic, no cancellation)
  * Gamma vs oracle, same constants ..... 1e-8   (Gamma = prefactor*(T(tau_in)-T(tau_out)) cancels: an
                                                  ulp of log10(tau) -- libm vs the kernel's log2 --
                                                  moves the table index by ~1e-12 and Gamma by
                                                  ~1e-12*T/dT; the reference's own two paths differ
                                                  by the same mechanism)
  * Gamma vs the reference Fortran golden, CUDA constants ... 1e-5 (documented ~1e-7 differences)
  * chemistry vs golden ................. 1e-9   (device exp/pow vs libm, amplified by |delth*dt|)
"""
import os

import numpy as np
import pytest

import cases
from oracle import oracle as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GAMMA_RTOL = 1e-8


@pytest.fixture(scope="module")
def asora():
    import pyc2ray_amd as p
    from pyc2ray_amd import _capi
    from pyc2ray_amd.load_extensions import load_asora
    lib = load_asora()
    yield p, lib, _capi
    if p.cuda_is_init():
        p.device_close()


def _setup(p, lib, c, N):
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(c["thin"], c["thick"])
    pos0, flux = cases.flat_sources(c["pos"], c["flux"])
    lib.source_data_to_device(pos0, flux, flux.shape[0])
    lib.density_to_device(np.ravel(c["ndens"]).astype("float64", copy=True), N)
    return pos0, flux


def _asora_call(lib, c, N, numtau):
    xh_flat = np.ravel(c["xh"]).astype("float64", copy=True)
    phi_flat = np.ravel(np.zeros((N, N, N)))
    cd_flat = np.ravel(np.zeros((N, N, N)))
    nd_flat = np.ravel(c["ndens"]).astype("float64", copy=True)
    r = lib.do_all_sources(c["R"], cd_flat, c["sig"], c["dr"], nd_flat, xh_flat, phi_flat, c["flux"].shape[0], N,
                           c["minlogtau"], c["dlogtau"], numtau)
    assert r is None
    return phi_flat.reshape(N, N, N)


@pytest.mark.parametrize("name", list(cases.RT_CASES))
@pytest.mark.parametrize("tables", ["grey", "soft"])
def test_raytrace_matches_oracle_and_reference(asora, name, tables):
    p, lib, capi = asora
    c = cases.rt_case(name, tables)
    N = c["N"]
    numtau = c["thin"].shape[0] - 1
    pos0, flux = _setup(p, lib, c, N)
    g = np.load(os.path.join(G, "raytrace.npz"))
    gold = g[f"{name}__{tables}__phi"]

    # default constants = the CUDA library being replaced
    lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 0)
    phi = _asora_call(lib, c, N, numtau)
    ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0, flux, c["thin"], c["thick"],
                                 c["minlogtau"], c["dlogtau"], NumTau=numtau, flags=O.ASORA_MODE)
    np.testing.assert_allclose(phi, ref["phi_ion"], rtol=GAMMA_RTOL, atol=0)
    np.testing.assert_allclose(phi, gold, rtol=1e-5, atol=0)            # north-star bar vs the Fortran
    gam, ev = lib.last_raytrace_counts()
    assert gam == int((ref["phi_ion"] != 0).sum()) or flux.shape[0] > 1
    assert ev >= gam

    # Fortran constants: agrees with the reference Fortran output to rounding
    lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 1)
    phi_f = _asora_call(lib, c, N, numtau)
    lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 0)
    np.testing.assert_allclose(phi_f, gold, rtol=GAMMA_RTOL, atol=0)


@pytest.mark.parametrize("name", ["u16_1src_R8", "l16_7src_R5.5", "l17_3src_Rbox", "l32_5src_R10"])
def test_column_density_matches_oracle_and_reference(asora, name):
    p, lib, capi = asora
    c = cases.rt_case(name, "grey")
    N = c["N"]
    pos0, flux = _setup(p, lib, c, N)
    lib.grid_to_device(capi.GRID_XH_AV, c["xh"])
    last = flux.shape[0] - 1
    cd = lib.debug_coldens(c["R"], c["sig"], c["dr"], last, N)
    ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0, flux, c["thin"], c["thick"],
                                 c["minlogtau"], c["dlogtau"], NumTau=c["thin"].shape[0] - 1, flags=O.ASORA_MODE,
                                 want_coldens=True)["coldens"]
    w = cd != 0
    assert w.sum() > 0
    np.testing.assert_allclose(cd[w], ref[w], rtol=1e-12)
    # every cell that received a rate from this source has a column density
    only = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0[3 * last:], flux[last:],
                                  c["thin"], c["thick"], c["minlogtau"], c["dlogtau"],
                                  NumTau=c["thin"].shape[0] - 1, flags=O.ASORA_MODE)["phi_ion"]
    assert np.array_equal(w, only != 0)
    # and against the reference Fortran's scratch of its last source (sqrt literals differ by 1.8e-8)
    gold = np.load(os.path.join(G, "raytrace.npz"))[f"{name}__grey__cd"]
    np.testing.assert_allclose(cd[w], gold[w], rtol=1e-5)


def test_numtau_equal_table_length_is_clamped(asora):
    """evolve3D passes NumTau = len(table) (pyc2ray/evolve.py:124); the reference then reads one
    element past the table for tau >= 10^maxlogtau.  This build clamps; the oracle does the same."""
    p, lib, capi = asora
    c = cases.rt_case("l16_thick", "soft")
    N = c["N"]
    pos0, flux = _setup(p, lib, c, N)
    L = c["thin"].shape[0]
    phi = _asora_call(lib, c, N, L)
    ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0, flux, c["thin"], c["thick"],
                                 c["minlogtau"], c["dlogtau"], NumTau=L, flags=O.ASORA_MODE)["phi_ion"]
    assert np.isfinite(phi).all()
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


def test_z_transposed_layout_is_only_a_layout(asora):
    p, lib, capi = asora
    c = cases.rt_case("l32_5src_R10", "soft")
    N = c["N"]
    _setup(p, lib, c, N)
    lib.set_option(capi.OPT_Z_TRANSPOSED, 1)
    a = _asora_call(lib, c, N, c["thin"].shape[0] - 1)
    lib.set_option(capi.OPT_Z_TRANSPOSED, 0)
    b = _asora_call(lib, c, N, c["thin"].shape[0] - 1)
    lib.set_option(capi.OPT_Z_TRANSPOSED, 1)
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", ["u16_1src_R8", "l16_7src_R5.5", "l17_3src_Rbox", "l32_5src_R10", "l16_thick"])
@pytest.mark.parametrize("threads", [64, 128, 256, 512, 1024])
def test_decomposition_and_workgroup_size_do_not_change_results(asora, name, threads):
    """One workgroup per octant vs one per (octant, sector), per mirrored sector pair, per quarter sector and per mirrored pair
    of whole octants, at every
    workgroup size: same Gamma and the same count of rated pairs; the sector forms evaluate more column densities
    (re-derived planes; the quarter sectors re-derive what feeds them)."""
    p, lib, capi = asora
    c = cases.rt_case(name, "soft")
    N = c["N"]
    pos0, flux = _setup(p, lib, c, N)
    numtau = c["thin"].shape[0] - 1
    ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0, flux, c["thin"], c["thick"],
                                 c["minlogtau"], c["dlogtau"], NumTau=numtau, flags=O.ASORA_MODE)["phi_ion"]
    out = {}
    try:
        lib.set_option(capi.OPT_BLOCK_THREADS, threads)
        for mode in (1, 2, 3, 4, 5, 6, 7, 8, 9):
            lib.set_option(capi.OPT_SECTORS, mode)
            phi = _asora_call(lib, c, N, numtau)
            np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)
            out[mode] = (phi, lib.last_raytrace_counts())
    finally:
        lib.set_option(capi.OPT_SECTORS, 0)
        lib.set_option(capi.OPT_BLOCK_THREADS, 0)
    np.testing.assert_allclose(out[1][0], out[2][0], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out[1][0], out[3][0], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out[1][0], out[4][0], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out[1][0], out[5][0], rtol=1e-12, atol=0)
    for mode in (6, 7, 8, 9):                         # whole sphere, half spheres, all-sign sectors, sectors by dominant sign
        np.testing.assert_allclose(out[1][0], out[mode][0], rtol=1e-12, atol=0)
    assert len({out[mode][1][0] for mode in out}) == 1                                     # rated pairs
    assert out[6][1][1] == out[6][1][0]               # the whole sphere in one workgroup: nothing is evaluated twice
    assert out[6][1][1] <= out[7][1][1] <= out[5][1][1] <= out[1][1][1]
    assert out[2][1][1] >= out[1][1][1]               # evaluations (re-derived planes)
    assert out[4][1][1] >= out[2][1][1]               # quarter sectors re-derive the inner part of their sector


def test_grey_notables_option(asora):
    p, lib, capi = asora
    c = cases.rt_case("l16_7src_R5.5", "grey")
    N = c["N"]
    pos0, flux = _setup(p, lib, c, N)
    lib.set_option(capi.OPT_GREY_NOTABLES, 1)
    try:
        phi = _asora_call(lib, c, N, c["thin"].shape[0] - 1)
    finally:
        lib.set_option(capi.OPT_GREY_NOTABLES, 0)
    ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], c["ndens"], c["xh"], pos0, flux, c["thin"], c["thick"],
                                 c["minlogtau"], c["dlogtau"], flags=O.ASORA_MODE | O.GREY)["phi_ion"]
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


@pytest.mark.parametrize("N,R", [(48, 1000.0), (40, 17.3)])
def test_large_radius_and_window_clipping(asora, N, R):
    """R beyond the box: the trace is cut by the periodic window (raytracing.cu:122-123,241) and by
    q_max (raytracing.cu:101).  N=48 full box needs 6*25^2*8 B = 30 KB of LDS shells."""
    p, lib, capi = asora
    nd, xh, dr = cases.grid(N, "lognormal", 31, 0.04)
    pos, flux = cases.sources(N, 2, 32, flux=2.0)
    thin, thick, dlog = cases.soft_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, R=R, thin=thin, thick=thick, dlogtau=dlog,
             minlogtau=cases.MINLOGTAU, sig=cases.SIG)
    pos0, fl = _setup(p, lib, c, N)
    phi = _asora_call(lib, c, N, thin.shape[0] - 1)
    ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, pos0, fl, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0] - 1, flags=O.ASORA_MODE)["phi_ion"]
    assert (ref != 0).sum() == (phi != 0).sum()
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5, 0])
def test_large_shells_global_scratch_and_large_lds(asora, mode):
    """N=168 full box.  One workgroup per octant (mode 1): shell buffers 2*21.8k*8 B = 349 KB > 160 KB of LDS
    -> the global-scratch variant.  One per (octant, sector) (mode 2, what the library picks at this size):
    117 KB of dynamic LDS per workgroup."""
    p, lib, capi = asora
    N = 168
    nd, xh, dr = cases.grid(N, "lognormal", 41, 0.02)
    pos, flux = cases.sources(N, 1, 42, flux=5.0)
    thin, thick, dlog = cases.grey_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, R=1000.0, thin=thin, thick=thick, dlogtau=dlog,
             minlogtau=cases.MINLOGTAU, sig=cases.SIG)
    pos0, fl = _setup(p, lib, c, N)
    lib.set_option(capi.OPT_SECTORS, mode)
    try:
        phi = _asora_call(lib, c, N, thin.shape[0] - 1)
    finally:
        lib.set_option(capi.OPT_SECTORS, 0)
    ref = O.asora_do_all_sources(1000.0, cases.SIG, dr, nd, xh, pos0, fl, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0] - 1, flags=O.ASORA_MODE)["phi_ion"]
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


# ---- chemistry -------------------------------------------------------------------------------
@pytest.mark.parametrize("N,seed", [(16, 21), (12, 22)])
@pytest.mark.parametrize("order", ["F", "C"])
def test_global_pass_matches_reference(asora, N, seed, order):
    from pyc2ray_amd.load_extensions import load_c2ray
    g = np.load(os.path.join(G, "global_pass.npz"))
    c = cases.chem_case(N, seed)
    mk = lambda a: np.array(a, order=order, copy=True)
    xh_av, xh_int, xh0 = mk(c["xh_av"]), mk(c["xh_intermed"]), mk(c["xh"])
    conv = load_c2ray().chemistry.global_pass(c["dt"], mk(c["ndens"]), mk(c["temp"]), xh0, xh_av, xh_int,
                                              mk(c["phi_ion"]), c["bh00"], c["albpow"], c["colh0"], c["temph0"],
                                              c["abu_c"])
    np.testing.assert_allclose(xh_av, g[f"n{N}_xh_av"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(xh_int, g[f"n{N}_xh_intermed"], rtol=1e-9, atol=0)
    assert conv == int(g[f"n{N}_conv"])
    assert np.array_equal(xh0, c["xh"])                       # xh itself is not modified


def test_chemistry_tutorial_known_answer_on_gpu(asora):
    import pyc2ray_amd as p
    mesh = (10, 10, 10)
    np.random.seed(2023)
    ndens = np.random.normal(loc=1e-7, scale=1e-8, size=mesh)
    temp = np.ones(mesh) * 1e4
    xh = np.random.uniform(low=0, high=0.1, size=mesh)
    phi = np.random.uniform(low=1e-13, high=1e-12, size=mesh)
    assert round(xh.mean(), 3) == 0.050
    for _ in range(100):
        xh = p.hydrogenODE(dt=50 * 3.15576e7, ndens=ndens, temp=temp, xh=xh, phi_ion=phi)
    assert round(xh.mean(), 3) == 0.127                       # tutorials/chemistry_solver.ipynb cell 5


# ---- the whole step --------------------------------------------------------------------------
@pytest.mark.parametrize("order", ["F", "C"])
def test_evolve3D_matches_oracle_loop(asora, order, tmp_path):
    from evolve_oracle import evolve3D_oracle
    p, lib, capi = asora
    N = 24
    nd, xh, dr = cases.grid(N, "lognormal", 51, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, 4, 52, flux=30.0)
    thin, thick, dlog = cases.soft_tables()
    mk = lambda a: np.array(a, order=order, copy=True)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    dt = 3.15576e13 * 5
    args = dict(dt=dt, dr=dr, src_flux=flux, src_pos=pos, use_gpu=True, max_subbox=1000, subboxsize=N,
                loss_fraction=1e-2, temp=mk(temp), ndens=mk(nd), xh=mk(xh), photo_thin_table=thin,
                photo_thick_table=thick, minlogtau=cases.MINLOGTAU, dlogtau=dlog, R_max_LLS=9.0,
                convergence_fraction=1e-4, sig=cases.SIG, bh00=cases.BH00, albpow=cases.ALBPOW, colh0=cases.COLH0,
                temph0=cases.TEMPH0, abu_c=cases.ABU_C, logfile=str(tmp_path / "log.txt"), quiet=True)
    xh_new, phi = p.evolve3D(**args)
    niter = p.evolve._evolve.last_niter
    x_ref, phi_ref, niter_ref, hist = evolve3D_oracle(dt, dr, flux, pos, temp, nd, xh, thin, thick, cases.MINLOGTAU,
                                                      dlog, 9.0, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW,
                                                      cases.COLH0, cases.TEMPH0, cases.ABU_C)
    assert niter == niter_ref and niter >= 2
    assert xh_new.shape == (N, N, N) and xh_new.flags[f"{order}_CONTIGUOUS"]
    np.testing.assert_allclose(xh_new, x_ref, rtol=1e-8, atol=0)
    np.testing.assert_allclose(phi, phi_ref, rtol=1e-7, atol=0)
    assert xh_new.max() > 0.5                                  # the sources did ionise their surroundings
    log = open(tmp_path / "log.txt").read()
    assert "Multiple source convergence reached." in log
    # the two means of the reference's log line are summed on the device: same printed digits as numpy's
    assert f"Mean density (cgs): {nd.mean():.3e}, Mean ionized fraction: {xh.mean():.3e}" in log


def test_grid_sum_and_page_locked_results(asora, tmp_path):
    """asora_grid_sum against numpy, and the arrays evolve3D returns: page-locked buffers of the library's pool
    (pyc2ray_amd/_pinned.py), recycled once the caller has dropped them, never while a view is alive."""
    import gc
    from pyc2ray_amd import _pinned
    p, lib, capi = asora
    N = 40
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    rng = np.random.default_rng(77)
    a = rng.lognormal(0.0, 2.0, (N, N, N))
    lib.grid_to_device(capi.GRID_NDENS, a)
    np.testing.assert_allclose(lib.grid_sum(capi.GRID_NDENS), a.sum(), rtol=1e-13)
    lib.grid_to_device(capi.GRID_XH, np.asfortranarray(-a))
    np.testing.assert_allclose(lib.grid_sum(capi.GRID_XH), -a.sum(), rtol=1e-13)
    with pytest.raises(RuntimeError):
        lib.grid_sum(capi.GRID_PHI_HEAT)                        # nothing was ever put there

    def owner(arr):
        while isinstance(arr, np.ndarray):
            arr = arr.base
        return arr

    thin, thick, dlog = cases.soft_tables()
    p.photo_table_to_device(thin, thick)
    nd, xh, dr = cases.grid(N, "lognormal", 5, 0.15, xlo=1e-4, xhi=2e-3)
    pos, flux = cases.sources(N, 3, 6, flux=30.0)
    temp = np.full((N, N, N), 1e4)
    step = lambda x: p.evolve3D(3.15576e13, dr, flux, pos, True, 1000, N, 1e-2, temp, nd, x, thin, thick, cases.MINLOGTAU, dlog,
                                9.0, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C,
                                logfile=str(tmp_path / "log.txt"), quiet=True)
    x1, phi1 = step(xh)
    assert isinstance(owner(x1), _pinned._Owner) and isinstance(owner(phi1), _pinned._Owner)
    assert _pinned.stats()["pinned_bytes"] >= 2 * N ** 3 * 8
    keep, view = x1.copy(), x1[3]
    addr_x1, addr_phi1 = x1.ctypes.data, phi1.ctypes.data
    x2, phi2 = step(x1)
    assert len({addr_x1, addr_phi1, x2.ctypes.data, phi2.ctypes.data}) == 4         # all four alive: four buffers
    np.testing.assert_array_equal(x1, keep)                                        # and the first result is untouched
    del x1, phi1
    gc.collect()
    x3, phi3 = step(x2)
    assert phi3.ctypes.data == addr_phi1 or x3.ctypes.data == addr_phi1            # the dropped rate grid's buffer is back in use
    assert addr_x1 not in (x3.ctypes.data, phi3.ctypes.data)                       # `view` still holds the other one
    np.testing.assert_array_equal(view, keep[3])
    assert x3.mean() > x2.mean() > keep.mean()


# ---- size-independent properties at the benchmark size ------------------------------------------
def test_full_size_properties_256(asora):
    """256^3, 64 sources, R=32: linearity in flux, superposition over sources, and the exact count of
    rate-receiving (source,cell) pairs."""
    p, lib, capi = asora
    N, Ns, R = 256, 64, 32.0
    nd, xh, dr = cases.grid(N, "lognormal", 61, 0.02)
    pos, flux = cases.sources(N, Ns, 100, flux=1.0)
    thin, thick, dlog = cases.soft_tables(20000)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 64)
    p.photo_table_to_device(thin, thick)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    numtau = thin.shape[0] - 1

    def trace(pos1, fl):
        p0, f0 = cases.flat_sources(pos1, fl)
        lib.source_data_to_device(p0, f0, f0.shape[0])
        lib.raytrace_device(R, cases.SIG, dr, 0, f0.shape[0], cases.MINLOGTAU, dlog, numtau)
        return lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))

    full = trace(pos, flux)
    gam, ev = lib.last_raytrace_counts()
    # number of lattice points with |d|^2 <= R^2 (R < N/2: no clipping)
    r = np.arange(-32, 33)
    inside = int(((r[:, None, None] ** 2 + r[None, :, None] ** 2 + r[None, None, :] ** 2) <= R * R).sum())
    assert gam == Ns * inside
    assert ev >= gam and ev < 1.15 * gam
    assert np.isfinite(full).all() and (full >= 0).all()
    # linearity: tripling every flux triples Gamma
    # (prefactor*T_in - prefactor*T_out rounds differently for 3x the prefactor: cancellation ~1e4 ulps)
    np.testing.assert_allclose(trace(pos, 3.0 * flux), 3.0 * full, rtol=1e-10, atol=0)
    # superposition: two halves of the source list add up (atomic summation order aside)
    a = trace(pos[:, :Ns // 2], flux[:Ns // 2])
    b = trace(pos[:, Ns // 2:], flux[Ns // 2:])
    np.testing.assert_allclose(a + b, full, rtol=1e-11, atol=0)
    # sub-range tracing equals tracing the sub-list
    p0, f0 = cases.flat_sources(pos, flux)
    lib.source_data_to_device(p0, f0, Ns)
    lib.raytrace_device(R, cases.SIG, dr, Ns // 2, Ns - Ns // 2, cases.MINLOGTAU, dlog, numtau)
    b2 = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
    np.testing.assert_allclose(b2, b, rtol=1e-11, atol=0)
    # one source of the list against the oracle on the full-size grid (R=32: ~0.1 s of CPU)
    ref1 = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, p0[:3], f0[:1], thin, thick, cases.MINLOGTAU, dlog,
                                  NumTau=numtau, flags=O.ASORA_MODE)["phi_ion"]
    np.testing.assert_allclose(trace(pos[:, :1], flux[:1]), ref1, rtol=GAMMA_RTOL, atol=0)


def test_uniform_medium_is_mirror_symmetric(asora):
    """Uniform density, one central source: Gamma is symmetric under every reflection through the
    source (the 8 octants are computed by 8 different workgroups)."""
    p, lib, capi = asora
    N = 64
    nd, xh, dr = cases.grid(N, "uniform", 0, 0.1)
    pos = np.array([[32], [32], [32]])
    flux = np.array([1e3])
    thin, thick, dlog = cases.grey_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, R=25.0, thin=thin, thick=thick, dlogtau=dlog,
             minlogtau=cases.MINLOGTAU, sig=cases.SIG)
    _setup(p, lib, c, N)
    phi = _asora_call(lib, c, N, thin.shape[0] - 1)
    s = 31                                                      # 0-based source index
    blk = phi[s - 25:s + 26, s - 25:s + 26, s - 25:s + 26]
    for ax in range(3):
        np.testing.assert_allclose(blk, np.flip(blk, axis=ax), rtol=1e-13, atol=0)
    # axis swaps change the order of a few sums (dist2, weights): equal to rounding, not bitwise
    np.testing.assert_allclose(blk, blk.transpose(1, 0, 2), rtol=1e-10, atol=0)
    np.testing.assert_allclose(blk, blk.transpose(2, 1, 0), rtol=1e-10, atol=0)


def test_error_behaviour(asora):
    p, lib, capi = asora
    if p.cuda_is_init():
        p.device_close()
    with pytest.raises(RuntimeError):
        p.device_close()                                        # asora_core.py:46-47
    with pytest.raises(RuntimeError):
        p.photo_table_to_device(np.ones(4), np.ones(4))         # asora_core.py:57
    p.device_init(16, 8)
    with pytest.raises(RuntimeError, match="outside the mesh"):
        lib.source_data_to_device(np.array([0, 0, 16], dtype=np.int32), np.ones(1), 1)
    with pytest.raises(TypeError, match="coldensh_out must be Array of type double"):
        lib.do_all_sources(4.0, np.zeros(16 ** 3, dtype=np.float32), 1.0, 1.0, np.zeros(16 ** 3), np.zeros(16 ** 3),
                           np.zeros(16 ** 3), 1, 16, -20.0, 0.01, 10)
    with pytest.raises(RuntimeError, match="does not match"):
        lib.density_to_device(np.zeros(8 ** 3), 8)
    p.device_close()


# ---- multi-GPU plumbing that can be checked on one GPU ----------------------------------------------
def test_rccl_allreduce_on_library_grid_world1(asora):
    """torch.distributed (backend nccl == RCCL) all-reduce applied in place to the library's
    device-resident phi_ion through a zero-copy __cuda_array_interface__ view.  One rank only (the box
    has one GPU): checks the aliasing, the stream hand-over and that RCCL accepts the pointer."""
    import socket
    import torch
    import torch.distributed as dist
    from pyc2ray_amd import dist as pd
    p, lib, capi = asora
    N = 32
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        pd.init_process_group_from_env("nccl")
    comm = pd.TorchComm()
    rng = np.random.default_rng(5)
    phi = rng.uniform(size=(N, N, N))
    lib.grid_to_device(capi.GRID_PHI_ION, phi)
    view = torch.as_tensor(pd._DevicePointer(lib.device_ptr(capi.GRID_PHI_ION), N ** 3), device="cuda")
    assert view.data_ptr() == lib.device_ptr(capi.GRID_PHI_ION)          # zero copy
    np.testing.assert_array_equal(view.cpu().numpy().reshape(N, N, N), phi)
    dist.all_reduce(view, op=dist.ReduceOp.SUM)                              # RCCL on the library's memory
    view.mul_(2.0)                                                           # write through the view
    torch.cuda.synchronize()
    back = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
    np.testing.assert_array_equal(back, 2.0 * phi)
    comm.allreduce_device_grid(lib, capi.GRID_PHI_ION, N)                    # world 1: no-op path
    p.device_close()


def test_slab_iteration_over_rccl_world1_equals_the_single_gpu_loop(asora, tmp_path):
    """evolve3D_MPI with a TorchComm over backend nccl (= RCCL), ONE rank (the box has one GPU): the slab path with its
    stream hand-over (torch ExternalStream on the library's stream), the zero-copy views of the library's grids, the chunked
    trace with folds, the slab chemistry and -- the part that needs RCCL -- the in-place all-reduce of the three
    convergence scalars on the library's reduction buffer, read back once per iteration.  Same iteration count and fields
    as evolve3D on the device-resident loop."""
    import socket
    import torch.distributed as dist
    from pyc2ray_amd import dist as pd
    p, lib, capi = asora
    N = 32
    nd, xh, dr = cases.grid(N, "lognormal", 91, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, 11, 92, flux=3e-4 * (N / 16.0) ** 3 / 11)
    thin, thick, dlog = cases.soft_tables()
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        pd.init_process_group_from_env("nccl")
    args = (3.15576e13 * 3, dr, flux, pos, True, 1000, N, 1e-2)
    rest = (temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, 7.0, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
            cases.TEMPH0, cases.ABU_C)
    x1, phi1 = p.evolve3D(*args, *rest, logfile=str(tmp_path / "a"), quiet=True)
    n1 = p.evolve._evolve.last_niter
    import pyc2ray_amd.evolve as E
    for chunks in (1, 3):
        comm = pd.TorchComm()
        comm.exchange, comm.slab_chunks = "slab", chunks
        # (one rank: evolve3D_MPI takes the distributed branch only with nprocs > 1; drive the slab path directly)
        from pyc2ray_amd.utils.sourceutils import format_sources
        spos, sflux, bounds = comm.shard_sources_by_slab(pos, flux, 1)
        plan = pd.SlabPlan(N, 1, 7.0, [spos[0] - 1])
        p0, f0 = format_sources(spos, sflux)
        lib.source_data_to_device(p0, f0, 11)
        for which, a in ((capi.GRID_NDENS, nd), (capi.GRID_TEMP, temp), (capi.GRID_XH, xh)):
            lib.grid_to_device(which, a)
        lib.grid_copy(capi.GRID_XH_AV, capi.GRID_XH)
        lib.grid_copy(capi.GRID_XH_INTERMED, capi.GRID_XH)
        chem = (3.15576e13 * 3, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
        ncell, prev1, prev0, niter, converged = N ** 3, 2 * N ** 3, 2 * N ** 3, 0, False
        crit = min(int(1e-4 * ncell), (11 - 1) / 3)
        while not converged:
            niter += 1
            conv, s1, s0 = comm.slab_iteration(lib, plan, N, 7.0, cases.SIG, dr, 11, cases.MINLOGTAU, dlog, thin.shape[0], chem, niter == 1)
            r1, r0 = abs((s1 - prev1) / s1), abs((s0 - prev0) / s0)
            converged = conv < crit or (r1 < 1e-4 and r0 < 1e-4)
            prev1, prev0 = s1, s0
            assert niter < 100
        x2 = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
        phi2 = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        assert niter == n1, (chunks, niter, n1)
        np.testing.assert_allclose(x2, x1, rtol=1e-10, atol=0)
        np.testing.assert_allclose(phi2, phi1, rtol=1e-10, atol=0)
        # the way evolve3D_MPI drives it: the convergence test on the device behind the in-place all-reduce, batches of 8
        # iterations per poll -- the launches enqueued beyond convergence must do nothing (same count, same fields)
        lib.grid_to_device(capi.GRID_XH, xh)
        comm.slab_begin(lib, plan, N, 7.0, cases.SIG, dr, 11, cases.MINLOGTAU, dlog, thin.shape[0], chem, crit, 1e-4)
        done, rows_all = False, []
        while not done:
            comm.slab_enqueue(lib, 8)
            n3, done, rows = comm.slab_poll(lib, 8)
            rows_all += list(rows)
            assert n3 < 100
        assert n3 == n1 == len(rows_all), (chunks, n3, n1, len(rows_all))
        comm.slab_enqueue(lib, 3)                                   # beyond convergence: nothing may change
        n4, done4, rows4 = comm.slab_poll(lib, 8)
        assert (n4, done4, len(rows4)) == (n1, True, 0)
        x3 = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
        phi3 = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        np.testing.assert_allclose(x3, x2, rtol=1e-10, atol=0)      # (the same kernels on the same inputs; the atomics' order is free)
        np.testing.assert_allclose(phi3, phi1, rtol=1e-10, atol=0)
        assert rows_all[-1][0] == conv and abs(rows_all[-1][1] - s1) <= 1e-10 * abs(s1)
    p.device_close()


def test_allreduce_loop_over_rccl_world1_equals_the_single_gpu_loop(asora, tmp_path, monkeypatch):
    """The reference's exchange (pyc2ray/evolve.py:433-437: the rate grid all-reduced, chemistry on identical data) on the
    device-resident loop (TorchComm.reduce_begin; asora_evolve_slab_fold_all): trace, both accumulator layouts folded into the
    out-box, the out-box all-reduced IN PLACE by RCCL on the library's stream (one rank here, forced), ONE fused pass on the
    whole grid reading the out-box, test on the device; batches of 8 iterations per poll.  Same iteration count and fields as
    evolve3D.  Iterations enqueued beyond convergence still all-reduce the (stale) out-box -- PHI_ION must not change."""
    import socket
    import torch.distributed as dist
    from pyc2ray_amd import dist as pd
    from pyc2ray_amd.utils.sourceutils import format_sources
    p, lib, capi = asora
    monkeypatch.setenv("PYC2RAY_AMD_FORCE_COLLECTIVE", "1")
    N = 32
    nd, xh, dr = cases.grid(N, "lognormal", 91, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, 11, 92, flux=3e-4 * (N / 16.0) ** 3 / 11)
    thin, thick, dlog = cases.soft_tables()
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        pd.init_process_group_from_env("nccl")
    args = (3.15576e13 * 3, dr, flux, pos, True, 1000, N, 1e-2)
    rest = (temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, 7.0, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
            cases.TEMPH0, cases.ABU_C)
    x1, phi1 = p.evolve3D(*args, *rest, logfile=str(tmp_path / "a"), quiet=True)
    n1 = p.evolve._evolve.last_niter
    comm = pd.TorchComm()
    comm.exchange = "allreduce"
    p0, f0 = format_sources(pos, flux)
    lib.source_data_to_device(p0, f0, 11)
    for which, a in ((capi.GRID_NDENS, nd), (capi.GRID_TEMP, temp), (capi.GRID_XH, xh)):
        lib.grid_to_device(which, a)
    chem = (3.15576e13 * 3, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    crit = min(int(1e-4 * N ** 3), (11 - 1) / 3)
    comm.phase_timing = True
    comm.reduce_begin(lib, N, 7.0, cases.SIG, dr, 11, cases.MINLOGTAU, dlog, thin.shape[0], chem, crit, 1e-4)
    done, rows_all = False, []
    while not done:
        comm.slab_enqueue(lib, 8)
        n2, done, rows = comm.slab_poll(lib, 8)
        rows_all += list(rows)
        assert n2 < 100
    assert n2 == n1 == len(rows_all), (n2, n1, len(rows_all))
    x2 = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
    phi2 = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
    np.testing.assert_allclose(x2, x1, rtol=1e-10, atol=0)
    np.testing.assert_allclose(phi2, phi1, rtol=1e-10, atol=0)
    assert np.array_equal(phi2 != 0, phi1 != 0)
    comm.slab_enqueue(lib, 3)                                   # beyond convergence
    n3, done3, rows3 = comm.slab_poll(lib, 8)
    assert (n3, done3, len(rows3)) == (n1, True, 0)
    assert np.array_equal(lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N))), phi2)
    assert np.array_equal(lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), x2)
    ph = comm.phase_report()
    assert set(ph) == {"trace_fold", "rate_allreduce", "pass_test", "iterations"} and ph["iterations"] >= n1
    # call order: a step that exchanges whole grids must own every plane, and folds before every pass
    with pytest.raises(RuntimeError, match="must own every plane"):
        lib.evolve_begin_slab(*chem, 7.0, cases.SIG, dr, cases.MINLOGTAU, dlog, thin.shape[0], 0, 11, -1.0, 0.0, 8, 8)
        lib.evolve_slab_fold_all()
    lib.evolve_begin_slab(*chem, 7.0, cases.SIG, dr, cases.MINLOGTAU, dlog, thin.shape[0], 0, 11, -1.0, 0.0, 0, N)
    lib.evolve_slab_trace(0, 11); lib.evolve_slab_fold_all(); lib.evolve_slab_pass(); lib.evolve_slab_close(None)
    lib.evolve_slab_trace(0, 11)
    with pytest.raises(RuntimeError, match="this iteration's fold has not been enqueued"):
        lib.evolve_slab_pass()
    lib.evolve_slab_fold_all(); lib.evolve_slab_pass(); lib.evolve_slab_close(None)
    assert lib.evolve_poll(4)[0] == 2
    # the out-box written from the host (transports that sum on the host)
    box = np.arange(2 * N * N, dtype=np.float64).reshape(2, N, N)
    lib.evolve_slab_outbox_from_host(5, box)
    assert np.array_equal(lib.evolve_slab_outbox_to_host(5, 2, N), box)
    p.device_close()


def test_grids_placed_by_probe_give_the_same_results(asora, tmp_path, monkeypatch):
    """device_init puts every N^3 grid of the hot loop into ONE allocation chosen among several by a streaming probe (api.hip
    choose_arena: placements differ by ~15 % in what the memory side delivers).  From 128^3 on several candidates are tried;
    ASORA_PLACEMENT_CANDIDATES=1 takes the first.  Whichever is kept, the grids are where the library thinks they are: a time step
    gives the same result either way, and re-initialising at another size leaves nothing behind."""
    p, lib, capi = asora
    N = 128
    nd, xh, dr = cases.grid(N, "lognormal", 23, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, 40, 24, flux=3e-4 * (N / 16.0) ** 3 / 40)
    thin, thick, dlog = cases.soft_tables()
    args = (3.15576e13 * 3, dr, flux, pos, True, 1000, N, 1e-2)
    rest = (temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, 9.0, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
            cases.TEMPH0, cases.ABU_C)
    out = {}
    for cand in ("1", "6"):
        monkeypatch.setenv("ASORA_PLACEMENT_CANDIDATES", cand)
        if p.cuda_is_init():
            p.device_close()
        p.device_init(N, 8)
        pl = lib.debug_placement()
        assert 1 <= pl["candidates"] <= int(cand), pl
        if cand == "1":
            assert pl["candidates"] == 1
        else:
            assert pl["candidates"] >= 2 and 0.0 < pl["chosen_probe_ms"] <= pl["slowest_probe_ms"], pl
        p.photo_table_to_device(thin, thick)
        x, phi = p.evolve3D(*args, *rest, logfile=str(tmp_path / cand), quiet=True)
        out[cand] = (x, phi, p.evolve._evolve.last_niter)
    assert out["1"][2] == out["6"][2]
    np.testing.assert_allclose(out["6"][0], out["1"][0], rtol=1e-10, atol=0)
    np.testing.assert_allclose(out["6"][1], out["1"][1], rtol=1e-10, atol=0)
    monkeypatch.delenv("ASORA_PLACEMENT_CANDIDATES")
    p.device_close()
    p.device_init(16, 8)                                   # small meshes: the first allocation
    assert lib.debug_placement()["candidates"] == 1
    p.device_close()


# ---- edge cases -----------------------------------------------------------------------------------------
def _edge_case(p, lib, capi, N, pos, flux, R, tau_cell=0.1, seed=71, tables="soft", xh_override=None):
    nd, xh, dr = cases.grid(N, "lognormal", seed, tau_cell)
    if xh_override is not None:
        xh = xh_override(xh)
    thin, thick, dlog = cases.soft_tables() if tables == "soft" else cases.grey_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=np.asarray(pos), flux=np.asarray(flux, dtype=float), R=R, thin=thin,
             thick=thick, dlogtau=dlog, minlogtau=cases.MINLOGTAU, sig=cases.SIG)
    pos0, fl = _setup(p, lib, c, N)
    phi = _asora_call(lib, c, N, thin.shape[0] - 1)
    ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, pos0, fl, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0] - 1, flags=O.ASORA_MODE)
    return phi, ref["phi_ion"]


@pytest.mark.parametrize("R", [0.0, 0.5, 1.0, 1.5, 2.0 ** 0.5, 3.0 ** 0.5, 2.0])
def test_tiny_radii(asora, R):
    """R = 0: only the source cell is rated; R = 1, sqrt2, sqrt3: the 6 / 18 / 26 nearest cells join
    (cells exactly on the sphere are classified as the reference's floating-point test does)."""
    p, lib, capi = asora
    phi, ref = _edge_case(p, lib, capi, 12, [[4], [7], [9]], [2.0], R)
    assert (phi != 0).sum() == (ref != 0).sum()
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


def test_sources_on_the_box_corners_wrap_periodically(asora):
    p, lib, capi = asora
    N = 20
    pos = np.array([[1, N, 1, N], [1, N, N, 1], [1, N, 1, 1]])
    phi, ref = _edge_case(p, lib, capi, N, pos, [1.0, 2.0, 3.0, 4.0], 7.3)
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


def test_coincident_sources_add_up(asora):
    """Several sources in the same cell: every rate atomic of theirs collides."""
    p, lib, capi = asora
    pos = np.array([[6] * 9, [6] * 9, [6] * 9])
    phi9, ref9 = _edge_case(p, lib, capi, 16, pos, [1.5] * 9, 6.0)
    phi1, _ = _edge_case(p, lib, capi, 16, pos[:, :1], [1.5], 6.0)
    np.testing.assert_allclose(phi9, ref9, rtol=GAMMA_RTOL, atol=0)
    np.testing.assert_allclose(phi9, 9.0 * phi1, rtol=1e-12, atol=0)


def test_no_sources_gives_zero_rates(asora):
    p, lib, capi = asora
    N = 12
    nd, xh, dr = cases.grid(N, "uniform", 0, 0.1)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    thin, thick, dlog = cases.grey_tables()
    p.photo_table_to_device(thin, thick)
    lib.source_data_to_device(np.zeros(0, dtype=np.int32), np.zeros(0), 0)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    lib.raytrace_device(5.0, cases.SIG, dr, 0, 0, cases.MINLOGTAU, dlog, thin.shape[0] - 1)
    phi = lib.grid_to_host(capi.GRID_PHI_ION, np.full((N, N, N), 7.0))
    assert not phi.any()
    with pytest.raises(RuntimeError, match="outside the 0 uploaded sources"):
        lib.raytrace_device(5.0, cases.SIG, dr, 0, 1, cases.MINLOGTAU, dlog, thin.shape[0] - 1)


def test_column_density_cap_stops_the_rates(asora):
    """Beyond N_HI = 2e30 cm^-2 the reference adds no rate (raytracing.cu:15,315); the column density keeps
    growing.  A huge cell size pushes most of the box over the cap."""
    p, lib, capi = asora
    phi, ref = _edge_case(p, lib, capi, 16, [[8], [8], [8]], [1.0], 1000.0, tau_cell=3e12, tables="grey")
    assert 0 < (ref != 0).sum() < ref.size // 2
    assert np.array_equal(phi != 0, ref != 0)
    w = ref != 0
    np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL, atol=0)


def test_nearly_ionised_medium_thin_cells(asora):
    """x -> 1: optical depths per cell far below 1e-7, every cell takes the optically thin branch."""
    p, lib, capi = asora
    phi, ref = _edge_case(p, lib, capi, 16, [[3, 12], [5, 9], [14, 2]], [1.0, 4.0], 9.0, tau_cell=1e-3,
                          xh_override=lambda x: 1.0 - 1e-7 * (1.0 + x))
    np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0)


# ---- physics known-answer: the reference's Test 1 (Stroemgren sphere) ----------------------------------------
def test_stroemgren_sphere_expansion(asora, tmp_path):
    """BASELINE configs[1] / the reference's paper test 1 (test/paper_tests/test1_Ifront): one source of
    1e54 photons/s in a uniform medium n_H = 1.87e-4 cm^-3, T = 1e4 K, grey opacity, box 5e24 cm, 128^3 cells,
    ten steps of 50 Myr.  The ionisation-front radius (x = 0.5 along the +i axis, as make_plot.ipynb cell 7 finds
    it) must follow the analytic r_I(t) = r_S (1 - exp(-t/t_rec))^(1/3), r_S = 964.377 kpc, t_rec = 654.266 Myr
    (make_plot.ipynb cell 5).  The reference's own figure shows r_N/r_A within [0.985, 1.005] at 256^3."""
    p, lib, capi = asora
    N = 128
    kpc, myr = 3.086e21, 3.15576e13
    boxsize = 5e24
    dr = boxsize / N
    ndens = np.full((N, N, N), 1.87e-7 * (1 + 9.0) ** 3, order="F")
    xh = np.full((N, N, N), 1.2e-3, order="F")
    temp = np.full((N, N, N), 1e4, order="F")
    src_pos = np.array([[64], [64], [64]])
    src_flux = np.array([1e54 / 1e48])
    thin, thick, dlog = cases.grey_tables(20000)
    colh0 = 1.3e-8 * 0.83 * 1.0 / 13.598 ** 2
    temph0 = 13.598 / 8.617e-05
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 1)
    p.photo_table_to_device(thin, thick)
    R_max_LLS = 15.0 * N / 1.62022035
    r_S = ((3 * 1e54) / (4 * np.pi * 2.59e-13 * 1.87e-4 ** 2)) ** (1. / 3) / kpc
    t_rec = 1.0 / (2.59e-13 * 1.87e-4 * myr)
    assert abs(r_S - 964.377) < 1e-3 and abs(t_rec - 654.266) < 1e-3
    x_axis = (np.arange(N - 63) * dr) / kpc                       # distance from the source cell along +i
    ratios = []
    for step in range(1, 11):
        xh, phi = p.evolve3D(50 * myr, dr, src_flux, src_pos, True, 1000, 128, 1e-2, temp, ndens, xh, thin, thick,
                             cases.MINLOGTAU, dlog, R_max_LLS, 1e-4, cases.SIG, 2.59e-13, -0.7, colh0, temph0, 7.1e-7,
                             logfile=str(tmp_path / "log"), quiet=True)
        prof = xh[63:, 63, 63]
        front = np.interp(0.5, np.flip(prof), np.flip(x_axis))
        r_A = r_S * (1.0 - np.exp(-50.0 * step / t_rec)) ** (1. / 3)
        ratios.append(front / r_A)
    ratios = np.array(ratios)
    print("r_N/r_A per 50 Myr step:", np.round(ratios, 4))
    assert np.all(ratios[1:] > 0.975) and np.all(ratios[1:] < 1.01), ratios
    assert 0.3 < xh.mean() < 0.5                                  # ~ (4/3 pi r_I^3)/box: 0.41 at 500 Myr
    p.device_close()


# ---- photo-heating (extension; arithmetic of the Fortran path) -------------------------------------------------
@pytest.mark.parametrize("name", ["l16_7src_R5.5", "l17_3src_Rbox", "l32_5src_R10", "l16_thin"])
def test_heating_rates_match_fortran_path(asora, name, tmp_path):
    """do_raytracing with real heating tables returns phi_heat computed on the GPU.  The checker is the oracle's
    restatement of the Fortran CPU path (photorates.f90:118,124, raytracing.f90:532,537), itself bit-identical to
    the compiled reference (tests/test_oracle_vs_reference.py)."""
    p, lib, capi = asora
    c = cases.rt_case(name, "soft")
    N = c["N"]
    hthin, hthick = 2.1e-11 * c["thin"] * np.linspace(1.0, 3.0, c["thin"].shape[0]), 1.7e-11 * c["thick"]
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(c["thin"][:-1], c["thick"][:-1])         # tables of NumTau = len-1 points
    lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 1)
    try:
        phi, heat = p.do_raytracing(c["dr"], c["flux"], c["pos"], True, 1000, N, 0.0, c["ndens"], c["xh"],
                                    c["thin"][:-1], c["thick"][:-1], hthin[:-1], hthick[:-1], c["minlogtau"],
                                    c["dlogtau"], c["R"], c["sig"], logfile=str(tmp_path / "log"), quiet=True)
    finally:
        lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 0)
    ref = O.do_all_sources(c["flux"], c["pos"], max_subbox=1000, subboxsize=N, sig=c["sig"], dr=c["dr"],
                           ndens=c["ndens"], xh_av=c["xh"], loss_fraction=0.0, thin=c["thin"][:-1],
                           thick=c["thick"][:-1], minlogtau=c["minlogtau"], dlogtau=c["dlogtau"], R_max_LLS=c["R"],
                           heat_thin=hthin[:-1], heat_thick=hthick[:-1], NumTau=c["thin"].shape[0] - 1)
    assert heat is not None and heat.shape == (N, N, N)
    np.testing.assert_allclose(phi, ref["phi_ion"], rtol=GAMMA_RTOL, atol=0)
    np.testing.assert_allclose(heat, ref["phi_heat"], rtol=GAMMA_RTOL, atol=0)
    assert np.array_equal(heat != 0, phi != 0)
    # zero heating tables (what evolve3D passes, evolve.py:193): no heating grid
    phi2, heat2 = p.do_raytracing(c["dr"], c["flux"], c["pos"], True, 1000, N, 0.0, c["ndens"], c["xh"], c["thin"][:-1],
                                  c["thick"][:-1], np.zeros(5), np.zeros(5), c["minlogtau"], c["dlogtau"], c["R"],
                                  c["sig"], logfile=str(tmp_path / "log"), quiet=True)
    assert heat2 is None
    p.device_close()


def test_pipelined_raytrace_allreduce_world1(asora, monkeypatch):
    """The opt-in pipelined path (chunks of sources in order of their first coordinate, finished slabs of the
    rate grid summed over ranks on a second stream while the next chunk is traced): one rank only, with the
    collective forced, so that the chunking, the slab folds, the stream/event hand-over and RCCL on sub-ranges
    of the library's grid are exercised; the result must equal the plain raytrace."""
    import socket
    import torch.distributed as dist
    from pyc2ray_amd import dist as pd
    p, lib, capi = asora
    N = 48
    nd, xh, dr = cases.grid(N, "lognormal", 61, 0.12)
    pos, flux = cases.sources(N, 40, 62, flux=2.0)
    flux = flux * (1.0 + 0.1 * np.arange(40))
    thin, thick, dlog = cases.soft_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, thin=thin, thick=thick)
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        pd.init_process_group_from_env("nccl")
    monkeypatch.setenv("PYC2RAY_AMD_FORCE_COLLECTIVE", "1")
    numtau = thin.shape[0] - 1
    for R in (5.0, 9.5, 30.0):                     # slabs final early / late / only at the end
        _setup(p, lib, c, N)
        lib.grid_to_device(capi.GRID_XH_AV, xh)
        lib.raytrace_device(R, cases.SIG, dr, 0, 40, cases.MINLOGTAU, dlog, numtau)
        plain = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        for chunks in (8, 5, 1):
            comm = pd.TorchComm(overlap=True, chunks=chunks)
            spos, sflux = comm.sort_sources_for_overlap(pos, flux)
            p0, f0 = cases.flat_sources(spos, sflux)
            lib.source_data_to_device(p0, f0, 40)
            comm.raytrace_and_allreduce(lib, N, R, cases.SIG, dr, 40, cases.MINLOGTAU, dlog, numtau,
                                        src_i0=spos[0].astype(np.int64) - 1)
            lib.synchronize()
            piped = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
            np.testing.assert_allclose(piped, plain, rtol=1e-12, atol=1e-300)
            assert np.array_equal(piped != 0, plain != 0)
        with pytest.raises(ValueError, match="ascending"):
            comm.raytrace_and_allreduce(lib, N, R, cases.SIG, dr, 40, cases.MINLOGTAU, dlog, numtau,
                                        src_i0=(spos[0].astype(np.int64) - 1)[::-1])
    # ... the heating accumulator is folded slab by slab as well
    lib.heat_table_to_device(2e-11 * thin, 1e-11 * thick, thin.shape[0])
    lib.set_option(capi.OPT_HEATING, 1)
    try:
        _ = pos  # same sources
        comm = pd.TorchComm(overlap=True, chunks=6)
        spos, sflux = comm.sort_sources_for_overlap(pos, flux)
        p0, f0 = cases.flat_sources(spos, sflux)
        lib.source_data_to_device(p0, f0, 40)
        lib.raytrace_device(9.5, cases.SIG, dr, 0, 40, cases.MINLOGTAU, dlog, numtau)
        heat_plain = lib.grid_to_host(capi.GRID_PHI_HEAT, np.empty((N, N, N)))
        comm.raytrace_and_allreduce(lib, N, 9.5, cases.SIG, dr, 40, cases.MINLOGTAU, dlog, numtau,
                                    src_i0=spos[0].astype(np.int64) - 1)
        lib.synchronize()
        heat_piped = lib.grid_to_host(capi.GRID_PHI_HEAT, np.empty((N, N, N)))
        assert heat_plain.max() > 0
        np.testing.assert_allclose(heat_piped, heat_plain, rtol=1e-12, atol=1e-300)
    finally:
        lib.set_option(capi.OPT_HEATING, 0)
    # ... and with the chemistry pipelined behind each slab's sum: same grids, same convergence scalars
    chem = (3.15576e13, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    temp = np.full((N, N, N), 1e4)
    for chunks in (8, 3):
        res = {}
        for mode in ("plain", "piped", "piped_chem"):
            comm = pd.TorchComm(overlap=(mode != "plain"), chunks=chunks, pipeline_chemistry=(mode == "piped_chem"))
            spos, sflux = comm.sort_sources_for_overlap(pos, flux)
            p0, f0 = cases.flat_sources(spos, sflux)
            lib.source_data_to_device(p0, f0, 40)
            lib.grid_to_device(capi.GRID_TEMP, temp)
            lib.grid_to_device(capi.GRID_XH, xh)
            lib.grid_copy(capi.GRID_XH_AV, capi.GRID_XH)
            lib.grid_copy(capi.GRID_XH_INTERMED, capi.GRID_XH)
            scal = comm.raytrace_and_allreduce(lib, N, 9.5, cases.SIG, dr, 40, cases.MINLOGTAU, dlog, numtau,
                                               src_i0=spos[0].astype(np.int64) - 1, chemistry=chem)
            res[mode] = (scal, lib.grid_to_host(capi.GRID_XH_AV, np.empty((N, N, N))),
                         lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))))
        for mode in ("piped", "piped_chem"):
            assert res["plain"][0][0] == res[mode][0][0]                                   # conv_flag
            np.testing.assert_allclose(res["plain"][0][1:], res[mode][0][1:], rtol=1e-12)   # the two sums
            np.testing.assert_allclose(res[mode][1], res["plain"][1], rtol=1e-10)
            np.testing.assert_allclose(res[mode][2], res["plain"][2], rtol=1e-10)
    # the three-part C-ABI refuses to be used out of order
    p.device_close()
    p.device_init(N, 8)
    with pytest.raises(RuntimeError, match="no raytrace in progress"):
        lib.raytrace_range(0, 0)
    with pytest.raises(RuntimeError, match="no raytrace in progress"):
        lib.raytrace_fold(0, N)
    p.device_close()


def test_sharded_device_loop_refuses_calls_out_of_order_or_out_of_range(asora):
    """The asora_evolve_slab_* calls (include/asora_hip.h): no step begun, planes that are not the caller's to send or to receive,
    a close without a pass, the one-GPU enqueue on a sharded step -- error codes with messages, nothing launched."""
    p, lib, capi = asora
    N = 16
    nd, xh, dr = cases.grid(N, "lognormal", 5, 0.1)
    thin, thick, dlog = cases.soft_tables()
    pos, flux = cases.sources(N, 3, 6, flux=1.0)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    p0, f0 = cases.flat_sources(pos, flux)
    lib.source_data_to_device(p0, f0, 3)
    for which, a in ((capi.GRID_NDENS, nd), (capi.GRID_TEMP, np.full((N, N, N), 1e4)), (capi.GRID_XH, xh)):
        lib.grid_to_device(which, a)
    with pytest.raises(RuntimeError, match="no multi-GPU evolve step in progress"):
        lib.evolve_slab_trace(0, 3)
    chem = (3.15576e13, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    with pytest.raises(RuntimeError, match="bad range of own planes"):
        lib.evolve_begin_slab(*chem, 4.0, cases.SIG, dr, cases.MINLOGTAU, dlog, thin.shape[0], 0, 3, -1.0, 0.0, 12, 8)
    lib.evolve_begin_slab(*chem, 4.0, cases.SIG, dr, cases.MINLOGTAU, dlog, thin.shape[0], 0, 3, -1.0, 0.0, 8, 4)     # owns planes 8..11
    with pytest.raises(RuntimeError, match="begun with asora_evolve_begin_slab"):
        lib.evolve_enqueue(1)
    with pytest.raises(RuntimeError, match="outside the step's sources"):
        lib.evolve_slab_trace(2, 5)
    lib.evolve_slab_trace(0, 3)
    with pytest.raises(RuntimeError, match="planes this rank owns"):
        lib.evolve_slab_fold_out(10, 3)
    lib.evolve_slab_fold_out(12, 4)
    lib.evolve_slab_fold_out(0, 8)
    with pytest.raises(RuntimeError, match="does not own"):
        lib.evolve_slab_add_host(6, np.zeros((4, N, N)))
    lib.evolve_slab_add_host(8, np.zeros((2, N, N)))
    with pytest.raises(RuntimeError, match="pass has not been enqueued"):
        lib.evolve_slab_close((0, 1.0, 1.0))
    lib.evolve_slab_pass()
    with pytest.raises(RuntimeError, match="already enqueued"):
        lib.evolve_slab_pass()
    with pytest.raises(RuntimeError, match="pass has been enqueued already"):
        lib.evolve_slab_trace(0, 3)
    conv, s1, s0 = lib.chemistry_finish()                     # this rank's share of the three sums: its four planes
    assert 0 <= conv <= 4 * N * N and abs(s1 + s0 - 4 * N * N) < 1e-9 * 4 * N * N
    lib.evolve_slab_close((conv, s1, s0))
    n, done, rows = lib.evolve_poll(4)
    assert (n, done, len(rows)) == (1, False, 1) and rows[0][0] == conv
    p.device_close()


def test_beyond_the_last_table_entry_rates_are_exactly_zero(asora):
    """A medium so thick that the optical depth passes the last table entry (10^maxlogtau) a dozen cells from the
    source -- the reference benchmark's own medium does (tau = 228 per cell at 256^3).  Both table lookups then
    return the last entry and the reference's rate is EXACTLY 0; a fused pref*T_in - pref*T_out would leave the
    rounding error of one product there (random sign, ~1e-16 of the product).  Both GPU raytracers must give exact
    zeros on the oracle's zero set and no negative rate anywhere."""
    p, lib, capi = asora
    from pyc2ray_amd.load_extensions import load_c2ray
    N = 32
    nd, xh, dr = cases.grid(N, "uniform", 0, 900.0)
    pos, flux = cases.sources(N, 3, 71, flux=1.0)
    thin, thick, dlog = cases.soft_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, thin=thin, thick=thick)
    pos0, f0 = _setup(p, lib, c, N)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    numtau = thin.shape[0]
    ref = O.asora_do_all_sources(1000.0, cases.SIG, dr, nd, xh, pos0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=numtau,
                                 flags=O.ASORA_MODE)["phi_ion"]
    assert (ref == 0).sum() > 0.3 * N ** 3 and (ref > 0).sum() > 1000      # the case does what it is meant to
    for mode in (1, 2, 3):
        lib.set_option(capi.OPT_SECTORS, mode)
        lib.raytrace_device(1000.0, cases.SIG, dr, 0, 3, cases.MINLOGTAU, dlog, numtau)
        phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        assert not (phi < 0).any()
        assert np.array_equal(phi == 0, ref == 0)
        w = ref != 0
        np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL)
    lib.set_option(capi.OPT_SECTORS, 0)
    phi_f = np.zeros((N, N, N), order="F"); heat = np.zeros((N, N, N), order="F"); cd = np.zeros((N, N, N), order="F")
    load_c2ray().raytracing.do_all_sources(flux, pos, 1000, N, cd, cases.SIG, dr, nd, xh, phi_f, heat, 0.0, thin, thick,
                                           np.zeros(numtau), np.zeros(numtau), cases.MINLOGTAU, dlog, 1000.0)
    ref_f = O.do_all_sources(flux, pos, 1000, N, cases.SIG, dr, nd, xh, 0.0, thin, thick, cases.MINLOGTAU, dlog, 1000.0)["phi_ion"]
    assert not (phi_f < 0).any() and np.array_equal(phi_f == 0, ref_f == 0)


def test_infinite_optical_depth_reads_the_last_table_entry(asora):
    """A cell so dense that its outgoing optical depth overflows to +inf: the reference's log10(inf) = inf is clamped to the
    LAST table entry (min(NumTau, .), rates.cu:79 / photorates.f90:141).  The kernel's logarithm is built from the mantissa
    and exponent of its argument, which is only defined for finite values: the argument is clamped on both sides
    (rates_device.hpp).  Both sets of constants, every kind of unit, one and two sources per workgroup."""
    p, lib, capi = asora
    N = 16
    thin, thick, dlog = cases.soft_tables(400)
    nd = np.full((N, N, N), 1e-15)
    xh = np.full((N, N, N), 0.1)
    # (walls on the last shells inside the sphere: every cell that would read one lies beyond the radius.  A cell BEHIND such
    #  a wall is outside what either code is made for -- the reference's weight s/max(0.6, inf) is 0 there, the kernel's
    #  multiplied-through form of the same quotient has inf/inf -- and is not part of this test.)
    walls = [(8, 8, 14), (2, 8, 8), (8, 14, 8), (10, 3, 6)]
    for w in walls:
        nd[w] = 1e300                      # column density 1e300 x sigma 1e10 = +inf
    pos = np.array([[9, 9], [9, 9], [9, 9]])
    flux = np.array([2.0, 3.0])
    p0, f0 = cases.flat_sources(pos, flux)
    sig, dr, R = 1e10, 1.0, 6.0
    numtau = thin.shape[0]
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.source_data_to_device(p0, f0, 2)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    try:
        for fortran in (0, 1):
            flags = O.PER_SOURCE_FLUX if fortran else O.ASORA_MODE
            ref = O.asora_do_all_sources(R, sig, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=numtau,
                                         flags=flags)["phi_ion"]
            assert np.isfinite(ref).all() and all(ref[w] > 0 for w in walls)
            lib.set_option(capi.OPT_FORTRAN_CONSTANTS, fortran)
            for mode, pairs in ((0, 0), (1, 1), (3, 1), (6, 1), (6, 2), (9, 2)):
                lib.set_option(capi.OPT_SECTORS, mode)
                lib.set_option(capi.OPT_PAIR_SOURCES, pairs)
                lib.raytrace_device(R, sig, dr, 0, 2, cases.MINLOGTAU, dlog, numtau)
                phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
                assert np.isfinite(phi).all(), (fortran, mode, pairs)
                assert np.array_equal(phi != 0, ref != 0), (fortran, mode, pairs, np.argwhere((phi != 0) != (ref != 0))[:8].tolist())
                np.testing.assert_allclose(phi, ref, rtol=GAMMA_RTOL, atol=0, err_msg=str((fortran, mode, pairs)))
    finally:
        lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 0)
        lib.set_option(capi.OPT_SECTORS, 0)
        lib.set_option(capi.OPT_PAIR_SOURCES, 0)
    p.device_close()


def test_fully_ionised_and_empty_cells_behave_like_the_reference(asora):
    """Cells with nHI = 0 (x = 1 exactly, or no gas): the reference divides the cell's rate by nHI unguarded
    (raytracing.cu:324, raytracing.f90:531), so such a cell ends up with NaN while the column density passes through
    it unchanged and every other cell is unaffected.  Same here, in both GPU raytracers."""
    p, lib, capi = asora
    from pyc2ray_amd.load_extensions import load_c2ray
    c = cases.rt_case("l16_7src_R5.5", "soft")
    N = c["N"]
    xh = c["xh"].copy()
    nd = c["ndens"].copy()
    rng = np.random.default_rng(9)
    holes = rng.integers(0, N, size=(40, 3))
    for q, (i, j, k) in enumerate(holes):
        if q % 2:
            xh[i, j, k] = 1.0
        else:
            nd[i, j, k] = 0.0
    c = dict(c, xh=xh, ndens=nd, R=1000.0)
    pos0, flux = _setup(p, lib, c, N)
    numtau = c["thin"].shape[0] - 1
    with np.errstate(all="ignore"):
        ref = O.asora_do_all_sources(c["R"], c["sig"], c["dr"], nd, xh, pos0, flux, c["thin"], c["thick"], c["minlogtau"],
                                     c["dlogtau"], NumTau=numtau, flags=O.ASORA_MODE)["phi_ion"]
    assert np.isnan(ref).sum() >= 30 and np.isfinite(ref).sum() > 0.9 * N ** 3
    phi = _asora_call(lib, c, N, numtau)
    assert np.array_equal(np.isnan(phi), np.isnan(ref))
    ok = np.isfinite(ref)
    np.testing.assert_allclose(phi[ok], ref[ok], rtol=GAMMA_RTOL)
    phi_f = np.zeros((N, N, N), order="F"); heat = np.zeros((N, N, N), order="F"); cd = np.zeros((N, N, N), order="F")
    load_c2ray().raytracing.do_all_sources(c["flux"], c["pos"], 1000, N, cd, c["sig"], c["dr"], nd, xh, phi_f, heat, 0.0,
                                           c["thin"][:numtau], c["thick"][:numtau], np.zeros(numtau), np.zeros(numtau),
                                           c["minlogtau"], c["dlogtau"], 1000.0)
    with np.errstate(all="ignore"):
        r = O.do_all_sources(c["flux"], c["pos"], 1000, N, c["sig"], c["dr"], nd, xh, 0.0, c["thin"][:numtau],
                             c["thick"][:numtau], c["minlogtau"], c["dlogtau"], 1000.0)
    assert np.array_equal(np.isnan(phi_f), np.isnan(r["phi_ion"]))
    ok = np.isfinite(r["phi_ion"])
    np.testing.assert_allclose(phi_f[ok], r["phi_ion"][ok], rtol=1e-8, atol=1e-14 * np.nanmax(r["phi_ion"]))
    np.testing.assert_allclose(cd, r["coldens"], rtol=1e-11)


def test_randomised_asora_parameters_against_oracle(asora):
    """Seeded sweep of the ASORA path: mesh sizes (odd and even), radii from below one cell to beyond the box
    (integers included: cells exactly on the sphere), source counts with corner and duplicate positions, opacities
    from thin to beyond the last table entry, both NumTau conventions, both constant flavours."""
    p, lib, capi = asora
    rng = np.random.default_rng(4242)
    thin, thick, dlog = cases.soft_tables(500)
    for trial in range(30):
        N = int(rng.choice([9, 12, 16, 20, 24]))
        ns = int(rng.integers(1, 7))
        R = float(rng.choice([0.4, 1.0, 2.5, 3.0, 5.0, 7.3, N / 2, N * 0.9, 1000.0]))
        tau_cell = float(10 ** rng.uniform(-9.0, 3.2))
        kind = "lognormal" if trial % 3 else "uniform"
        nd, xh, dr = cases.grid(N, kind, 300 + trial, tau_cell)
        pos = 1 + rng.integers(0, N, size=(3, ns))
        if trial % 4 == 0:
            pos[:, 0] = [N, 1, N]
        if ns > 1 and trial % 5 == 0:
            pos[:, 1] = pos[:, 0]                               # two sources in one cell
        flux = rng.uniform(0.2, 5.0, size=ns)
        numtau = thin.shape[0] - (trial % 2)                    # len (evolve.py:124) or len-1 (the benchmark)
        fortran = (trial // 2) % 2
        c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, thin=thin, thick=thick)
        pos0, f0 = _setup(p, lib, c, N)
        lib.grid_to_device(capi.GRID_XH_AV, xh)
        lib.set_option(capi.OPT_FORTRAN_CONSTANTS, fortran)
        try:
            lib.raytrace_device(R, cases.SIG, dr, 0, ns, cases.MINLOGTAU, dlog, numtau)
            phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
            gam, ev = lib.last_raytrace_counts()
        finally:
            lib.set_option(capi.OPT_FORTRAN_CONSTANTS, 0)
        flags = O.PER_SOURCE_FLUX if fortran else O.ASORA_MODE
        ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, pos0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=numtau,
                                     flags=flags)
        tag = f"trial {trial}: N={N} ns={ns} R={R} tau={tau_cell:.3g} numtau={numtau} fortran={fortran}"
        assert np.array_equal(phi != 0, ref["phi_ion"] != 0), tag
        assert not (phi < 0).any(), tag
        scale = ref["phi_ion"].max()
        np.testing.assert_allclose(phi, ref["phi_ion"], rtol=GAMMA_RTOL, atol=1e-13 * scale, err_msg=tag)
        assert ev >= gam >= (ref["phi_ion"] != 0).sum(), tag


def test_randomised_chemistry_extremes_against_oracle(asora):
    """The fused chemistry kernel over a lattice of extreme inputs drawn cell by cell (rates from 0 to 1 s^-1,
    ionised fractions from 1e-14 to exactly 1, densities over nine decades, temperatures from 10 K to 1e6 K, time
    steps from a second to 1e18 s) against the oracle's global_pass: fields to 1e-9 (+1e-15 absolute), the
    count of non-converged cells to a handful (a cell sitting on one of the three thresholds may fall either way)."""
    from pyc2ray_amd.load_extensions import load_c2ray
    chem = load_c2ray().chemistry
    rng = np.random.default_rng(77)
    N = 14
    shape = (N, N, N)
    for trial, dt in enumerate([1.0, 3.15576e10, 3.15576e13, 1e15, 1e18]):
        phi = rng.choice([0.0, 1e-30, 1e-18, 1e-14, 1e-12, 1e-9, 1e-6, 1.0], size=shape)
        x0 = rng.choice([1e-14, 1e-6, 2e-4, 0.3, 1.0 - 1e-9, 1.0], size=shape)
        xav = np.clip(x0 * rng.uniform(0.3, 1.7, size=shape), 1e-14, 1.0)
        nd = rng.choice([1e-8, 1e-5, 1e-3, 0.1, 10.0], size=shape)
        temp = rng.choice([10.0, 1e3, 1e4, 5e4, 1e6], size=shape)
        xa_ref, xi_ref, conv_ref, _ = O.global_pass(dt, nd, temp, x0, xav, x0, phi, cases.BH00, cases.ALBPOW, cases.COLH0,
                                                    cases.TEMPH0, cases.ABU_C)
        xa, xi = xav.copy(), x0.copy()
        conv = chem.global_pass(dt, nd, temp, x0, xa, xi, phi, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0,
                                cases.ABU_C)
        # (atol: x = (x0 - x_eq) e^{-t/t_i} + x_eq cancels to ~1e-16 absolute when x0 << x_eq and t << t_i; the kernel
        #  forms it with one fused multiply-add, the reference with two roundings)
        np.testing.assert_allclose(xa, xa_ref, rtol=1e-9, atol=1e-15, err_msg=f"xh_av, dt={dt}")
        np.testing.assert_allclose(xi, xi_ref, rtol=1e-9, atol=1e-15, err_msg=f"xh_intermed, dt={dt}")
        assert abs(conv - conv_ref) <= 3, (dt, conv, conv_ref)
        assert np.isfinite(xa).all() and (xa >= 1e-14).all() and (xa <= 1.0).all()


# ---- the device-resident loop and the tiled chemistry pass ---------------------------------------------------------
@pytest.mark.parametrize("N", [17, 24])
def test_chemistry_slabs_equal_the_whole_pass_for_odd_and_even_meshes(asora, N):
    """asora_chemistry_range over uneven slabs (odd N: odd plane offsets) == asora_chemistry_device == the oracle."""
    p, lib, capi = asora
    c = cases.chem_case(N, 23 + N)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)

    def load():
        for which, key in ((capi.GRID_NDENS, "ndens"), (capi.GRID_TEMP, "temp"), (capi.GRID_XH, "xh"),
                           (capi.GRID_XH_AV, "xh_av"), (capi.GRID_XH_INTERMED, "xh_intermed"), (capi.GRID_PHI_ION, "phi_ion")):
            lib.grid_to_device(which, c[key])

    chem = (c["dt"], c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    load()
    conv, s1, s0 = lib.chemistry_device(*chem)
    xa = lib.grid_to_host(capi.GRID_XH_AV, np.empty((N, N, N)))
    xi = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
    load()
    cuts = [0, 1, 4, 9, N - 2, N]
    for q in range(len(cuts) - 1):
        lib.chemistry_range(*chem, cuts[q], cuts[q + 1] - cuts[q], q == 0)
    conv2, s1b, s0b = lib.chemistry_finish()
    assert conv2 == conv
    np.testing.assert_allclose([s1b, s0b], [s1, s0], rtol=1e-13)
    np.testing.assert_allclose(lib.grid_to_host(capi.GRID_XH_AV, np.empty((N, N, N))), xa, rtol=1e-13, atol=0)
    np.testing.assert_allclose(lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), xi, rtol=1e-13, atol=0)
    xa_ref, xi_ref, conv_ref, _ = O.global_pass(c["dt"], c["ndens"], c["temp"], c["xh"], c["xh_av"], c["xh_intermed"],
                                                c["phi_ion"], c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    assert abs(conv - conv_ref) <= 1
    np.testing.assert_allclose(xi, xi_ref, rtol=1e-9, atol=0)
    np.testing.assert_allclose(xa, xa_ref, rtol=1e-9, atol=0)


def test_chemistry_range_of_any_plane_count_fits_the_reduction_buffer(asora):
    """A range of planes can need MORE workgroups than the whole grid (the j-chunk count is rounded up per tile): at 256^3,
    65..96 planes ask for up to 4104 against 4096 -- the slab of a 3-rank run (85/86 planes).  Every such range must run,
    its three reductions must add up to the whole pass, and the rest of the grid must be left alone."""
    p, lib, capi = asora
    N = 256
    rng = np.random.default_rng(256)
    nd = 1e-3 * np.exp(0.5 * rng.standard_normal((N, N, N), dtype=np.float32).astype(np.float64))
    xh = np.full((N, N, N), 2e-4)
    temp = np.full((N, N, N), 1e4)
    phi = 1e-13 * rng.random((N, N, N), dtype=np.float32).astype(np.float64) ** 4
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    chem = (3.15576e13, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)

    def load():
        for which, a in ((capi.GRID_NDENS, nd), (capi.GRID_TEMP, temp), (capi.GRID_XH, xh), (capi.GRID_XH_AV, xh),
                         (capi.GRID_XH_INTERMED, xh), (capi.GRID_PHI_ION, phi)):
            lib.grid_to_device(which, a)

    load()
    conv, s1, s0 = lib.chemistry_device(*chem)
    xi = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
    assert conv > 1000
    for cuts in ([0, 85, 170, 256], [0, 65, 161, 256], [0, 96, 96 + 75, 256]):       # 3 ranks; 65, 96 and 75 planes
        load()
        for q in range(len(cuts) - 1):
            lib.chemistry_range(*chem, cuts[q], cuts[q + 1] - cuts[q], q == 0)
        conv2, s1b, s0b = lib.chemistry_finish()
        assert conv2 == conv
        np.testing.assert_allclose([s1b, s0b], [s1, s0], rtol=1e-13)
        assert np.array_equal(lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), xi)
    load()
    lib.chemistry_range(*chem, 100, 70, True)              # one range alone: the planes outside it keep their values
    part = lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N)))
    assert np.array_equal(part[100:170], xi[100:170]) and np.all(part[:100] == 2e-4) and np.all(part[170:] == 2e-4)
    p.device_close()


@pytest.mark.parametrize("N,ns,R", [(17, 3, 1000.0), (32, 5, 9.0), (40, 11, 13.5)])
def test_device_resident_loop_equals_the_step_by_step_loop(asora, N, ns, R, monkeypatch, tmp_path):
    """evolve3D on one GPU enqueues batches of outer iterations and lets the device evaluate the convergence test.
    Batch sizes 1, 3 and 8 must give identical results and the iteration count of the host-driven loop (raytrace_device
    + chemistry_device + the test of evolve.py:216-236 on the host), which is what the oracle loop restates."""
    from evolve_oracle import evolve3D_oracle
    import pyc2ray_amd.evolve as E
    p, lib, capi = asora
    nd, xh, dr = cases.grid(N, "lognormal", 70 + N, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, ns, 80 + N, flux=3e-4 * (N / 16.0) ** 3 / ns)
    thin, thick, dlog = cases.soft_tables()
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    dt = 3.15576e13 * 3
    args = (dt, dr, flux, pos, True, 1000, N, 1e-2, temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, R, 1e-4, cases.SIG,
            cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    results = []
    for batch in (1, 3, 8):
        monkeypatch.setattr(E, "EVOLVE_BATCH", batch)
        x, phi = p.evolve3D(*args, logfile=str(tmp_path / f"log{batch}"), quiet=True)
        results.append((x, phi, E._evolve.last_niter))
    for x, phi, niter in results[1:]:
        assert niter == results[0][2]
        np.testing.assert_allclose(x, results[0][0], rtol=1e-11, atol=0)         # the order the atomics add up in, only
        np.testing.assert_allclose(phi, results[0][1], rtol=1e-11, atol=0)
    x_ref, phi_ref, niter_ref, hist = evolve3D_oracle(dt, dr, flux, pos, temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog,
                                                      R, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
                                                      cases.TEMPH0, cases.ABU_C)
    assert results[0][2] == niter_ref and niter_ref >= 3
    np.testing.assert_allclose(results[0][0], x_ref, rtol=1e-8, atol=0)
    np.testing.assert_allclose(results[0][1], phi_ref, rtol=1e-7, atol=0)
    assert 0.02 < results[0][0].mean() < 0.98                                     # a partially ionised box
    # the log has one convergence line per iteration
    assert open(tmp_path / "log8").read().count("Number of non-converged points") == niter_ref


def test_fused_pass_skips_only_lines_no_source_reaches(asora):
    """Round 4: the fused pass neither reads nor zeroes the 64-byte lines of the rate accumulators that no source of the step can
    touch (State::reach_mask, built per (source upload, source range, R)).  A sequence of time steps ON ONE DEVICE STATE that
    walks through every transition -- same sources again (mask and accumulators reused as they stand), another source range, another
    radius, a radius that covers the box (mask off), back to a small one, a new upload, sources at the corners (wrap) -- each
    step through the device-resident loop against raytrace_device + chemistry_device called separately from the same start:
    iteration by iteration the same convergence numbers, at the end the same fields and rates.  A line left dirty by an
    earlier step, or a line of the sphere the mask misses, shows up as a wrong rate."""
    p, lib, capi = asora
    N = 40
    thin, thick, dlog = cases.soft_tables(600)
    nd, xh0, dr = cases.grid(N, "lognormal", 91, 0.4, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    rng = np.random.default_rng(92)
    numtau = thin.shape[0]
    chem = (3.15576e13 * 2, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_TEMP, temp)

    def upload(ns, corners=False):
        pos = 1 + rng.integers(0, N, size=(3, ns))
        if corners:
            pos[:, 0] = [1, 1, 1]; pos[:, 1] = [N, N, N]; pos[:, 2] = [1, N, 20]
        flux = rng.uniform(0.5, 2.0, size=ns) * 2e-3
        p0, f0 = cases.flat_sources(pos, flux)
        lib.source_data_to_device(p0, f0, ns)

    def separate(x_start, R, begin, count, iters):
        lib.grid_to_device(capi.GRID_XH, x_start)
        lib.grid_copy(capi.GRID_XH_AV, capi.GRID_XH)
        lib.grid_copy(capi.GRID_XH_INTERMED, capi.GRID_XH)
        rows = []
        for _ in range(iters):
            lib.raytrace_device(R, cases.SIG, dr, begin, count, cases.MINLOGTAU, dlog, numtau)
            rows.append(lib.chemistry_device(*chem))
        return (rows, lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N))))

    def fused(x_start, R, begin, count, iters):
        lib.grid_to_device(capi.GRID_XH, x_start)
        lib.evolve_begin(*chem, R, cases.SIG, dr, cases.MINLOGTAU, dlog, numtau, begin, count, -1.0, 0.0)
        lib.evolve_enqueue(iters)
        n_done, _, rows = lib.evolve_poll(iters)
        assert n_done == iters
        return (rows, lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N))))

    upload(9)
    x = xh0
    plan = [(3.0, 0, 9, 3, None), (3.0, 0, 9, 2, None),          # the same sources twice: everything reused
            (3.0, 2, 4, 3, None),                                # a sub-range: other lines
            (6.5, 0, 9, 2, None),                                # a larger radius
            (1000.0, 0, 9, 2, None),                             # the whole box: mask off
            (2.0, 0, 9, 3, None),                                # back to a small radius
            (4.0, 0, 5, 3, (5, True)),                           # new upload, sources on the corners (periodic wrap)
            (4.0, 0, 5, 1, None), (1.0, 1, 3, 2, None)]
    for step, (R, begin, count, iters, new_sources) in enumerate(plan):
        if new_sources:
            upload(*new_sources)
        f_rows, f_x, f_phi = fused(x, R, begin, count, iters)
        s_rows, s_x, s_phi = separate(x, R, begin, count, iters)
        tag = f"step {step}: R={R} sources [{begin},{begin + count}) iterations {iters}"
        for it in range(iters):
            assert int(f_rows[it][0]) == s_rows[it][0], tag
            np.testing.assert_allclose(f_rows[it][1:3], s_rows[it][1:3], rtol=1e-12, err_msg=tag)
        np.testing.assert_allclose(f_x, s_x, rtol=1e-10, atol=0, err_msg=tag)
        w = s_phi != 0
        assert np.array_equal(f_phi != 0, w), (tag, np.argwhere((f_phi != 0) != w)[:6].tolist())
        np.testing.assert_allclose(f_phi[w], s_phi[w], rtol=1e-10, atol=0, err_msg=tag)
        x = s_x
    p.device_close()


def test_uniform_temperature_form_of_the_chemistry_pass_is_bit_identical(asora):
    """A temperature grid found uniform at upload is not read again and its pow/sqrt/exp factors are evaluated once on
    the device; the general form (forced by ASORA_OPT_NO_UNIFORM_T) must give the same bits.  And a grid that is uniform
    except for ONE cell must take the general form by itself."""
    p, lib, capi = asora
    N = 24
    c = cases.chem_case(N, 77)
    chem = (c["dt"], c["bh00"], c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)

    def run(temp, force_general):
        for which, key in ((capi.GRID_NDENS, "ndens"), (capi.GRID_XH, "xh"), (capi.GRID_XH_AV, "xh_av"),
                           (capi.GRID_XH_INTERMED, "xh_intermed"), (capi.GRID_PHI_ION, "phi_ion")):
            lib.grid_to_device(which, c[key])
        lib.grid_to_device(capi.GRID_TEMP, temp)
        lib.set_option(capi.OPT_NO_UNIFORM_T, 1 if force_general else 0)
        try:
            lib.chemistry_range(*chem, 0, N, True)
            red = lib.chemistry_finish()
        finally:
            lib.set_option(capi.OPT_NO_UNIFORM_T, 0)
        return (lib.grid_to_host(capi.GRID_XH_AV, np.empty((N, N, N))), lib.grid_to_host(capi.GRID_XH_INTERMED, np.empty((N, N, N))), red)

    uniform = np.full((N, N, N), 8.7e3)
    a, b = run(uniform, False), run(uniform, True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    ref = O.global_pass(c["dt"], c["ndens"], uniform, c["xh"], c["xh_av"], c["xh_intermed"], c["phi_ion"], c["bh00"],
                        c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    np.testing.assert_allclose(a[1], ref[1], rtol=1e-9, atol=0)
    spoiled = uniform.copy()
    spoiled[N - 1, 3, 5] = 2.0e4
    s = run(spoiled, False)
    ref2 = O.global_pass(c["dt"], c["ndens"], spoiled, c["xh"], c["xh_av"], c["xh_intermed"], c["phi_ion"], c["bh00"],
                         c["albpow"], c["colh0"], c["temph0"], c["abu_c"])
    np.testing.assert_allclose(s[1], ref2[1], rtol=1e-9, atol=0)
    assert s[1][N - 1, 3, 5] != a[1][N - 1, 3, 5]


@pytest.mark.parametrize("N,R,ns", [(64, 5.5, 40), (72, 12.0, 25), (64, 3.0, 3)])
def test_pipelined_copies_of_the_drop_in_call_change_nothing(asora, N, R, ns):
    """libasora.do_all_sources overlaps its upload of xh_av, the trace and the download of phi_ion slab by slab (sources
    taken in order of their first coordinate).  Same rates as the plain upload-trace-download sequence (up to the order
    of the atomic sums) and as the oracle; sources near the periodic seam, empty slabs of sources, a second call on
    the same buffers."""
    p, lib, capi = asora
    nd, xh, dr = cases.grid(N, "lognormal", 90 + N, 0.1)
    pos, flux = cases.sources(N, ns, 91 + N, flux=2.0)
    pos[0, :3] = [1, N, N - 1]                                   # sources on the first and last planes: reach wraps
    flux = flux * (1.0 + 0.1 * np.arange(ns))
    thin, thick, dlog = cases.soft_tables()
    c = dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, R=R, thin=thin, thick=thick, dlogtau=dlog,
             minlogtau=cases.MINLOGTAU, sig=cases.SIG)
    p0, f0 = _setup(p, lib, c, N)
    numtau = thin.shape[0] - 1
    assert lib.get_option(capi.OPT_PIPELINED_COPIES) == 1
    piped = _asora_call(lib, c, N, numtau)
    gam = lib.last_raytrace_counts()[0]
    piped2 = _asora_call(lib, c, N, numtau)
    lib.set_option(capi.OPT_PIPELINED_COPIES, 0)
    try:
        plain = _asora_call(lib, c, N, numtau)
    finally:
        lib.set_option(capi.OPT_PIPELINED_COPIES, 1)
    assert lib.last_raytrace_counts()[0] == gam
    np.testing.assert_allclose(piped, plain, rtol=1e-12, atol=0)
    np.testing.assert_allclose(piped2, plain, rtol=1e-12, atol=0)
    ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=numtau,
                                 flags=O.ASORA_MODE)["phi_ion"]
    np.testing.assert_allclose(piped, ref, rtol=GAMMA_RTOL, atol=0)
    # the device-resident xh_av the call leaves behind is the uploaded one
    assert np.array_equal(lib.grid_to_host(capi.GRID_XH_AV, np.empty((N, N, N))), xh)


def test_leaving_out_exactly_zero_rates_changes_nothing(asora):
    """A thick cell whose optical depth lies beyond the last table entry gets pref * (T_last - T_last) = +0.  With
    ASORA_OPT_SKIP_ZERO_RATES = 0 (default) the buffer-atomic kernels do not add it (no atomic; waves with nothing to add skip
    the rate arithmetic), with 1 every kernel leaves it out, with 2 every rated cell is looked up and added, as the reference
    does.  One source (no summation-order freedom): the three grids must be IDENTICAL, zeros included, in an optically thick
    medium where most cells are beyond the table, in a thin one where none is, and with NumTau given as the table length and
    as length - 1; the library's count of cells left out is what the grid says."""
    p, lib, capi = asora
    N = 48
    thin, thick, dlog = cases.soft_tables(400)
    pos = np.array([[20], [31], [7]])
    flux = np.array([2.5])
    for tau_cell in (3000.0, 3.0, 1e-3):
        nd, xh, dr = cases.grid(N, "lognormal", 5, tau_cell, xlo=1e-4, xhi=1e-3)
        if p.cuda_is_init():
            p.device_close()
        p.device_init(N, 8)
        p.photo_table_to_device(thin, thick)
        p0, f0 = cases.flat_sources(pos, flux)
        lib.source_data_to_device(p0, f0, 1)
        lib.grid_to_device(capi.GRID_NDENS, nd)
        lib.grid_to_device(capi.GRID_XH_AV, xh)
        for numtau in (thin.shape[0], thin.shape[0] - 1):
            out, left_out = [], []
            for skip, atomics in ((1, 0), (0, 0), (2, 0), (1, 1), (0, 1)):
                lib.set_option(capi.OPT_SKIP_ZERO_RATES, skip)
                lib.set_option(capi.OPT_GLOBAL_ATOMICS, atomics)      # 1: the kernels without buffer atomics
                try:
                    lib.raytrace_device(1000.0, cases.SIG, dr, 0, 1, cases.MINLOGTAU, dlog, numtau)
                    left_out.append(lib.last_raytrace_zero_rates())
                finally:
                    lib.set_option(capi.OPT_SKIP_ZERO_RATES, 0)
                    lib.set_option(capi.OPT_GLOBAL_ATOMICS, 0)
                out.append(lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N))))
            for o in out[1:]:
                assert np.array_equal(out[0], o)
            # whole box traced, every cell rated once: what was left out is zero in the grid
            assert left_out[2] == 0 and left_out[4] == 0
            assert left_out[0] == left_out[1] == left_out[3]
            assert left_out[0] <= int((out[0] == 0).sum())
            assert (left_out[0] > 0.3 * N ** 3) if tau_cell == 3000.0 else (left_out[0] == 0)
            ref = O.asora_do_all_sources(1000.0, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog,
                                         NumTau=numtau, flags=O.ASORA_MODE)["phi_ion"]
            assert np.array_equal(out[0] == 0, ref == 0)
            w = ref != 0
            np.testing.assert_allclose(out[0][w], ref[w], rtol=GAMMA_RTOL, atol=0)
        zeros = float((out[0] == 0).mean())
        assert (zeros > 0.5) if tau_cell == 3000.0 else (zeros < 0.5)


def test_cells_on_the_sphere_follow_dr_without_a_rebuild(asora):
    """Integer radii have lattice points exactly ON the sphere (R = 5: (3,4,0), (5,0,0), ...; R = 9: (1,4,8), (4,4,7), ...).
    Whether such a cell is rated is the reference's floating-point test dist2/(dr*dr) <= R*R, whose outcome depends on dr.
    The tables keep those cells and re-decide their RATE bit in place when only dr changes (every step of a cosmological
    run): a trace after a change of dr must equal the oracle's and, bit for bit, the trace of a library that built its
    tables for that dr from scratch -- for every decomposition."""
    p, lib, capi = asora
    N = 32
    thin, thick, dlog = cases.soft_tables(400)
    nd, xh, dr0 = cases.grid(N, "lognormal", 21, 0.05, xlo=1e-4, xhi=1e-2)
    pos, flux = cases.sources(N, 3, 22, flux=2.0)
    pos[:, 0] = [16, 16, 16]
    pos[:, 1] = [3, 30, 16]
    p0, f0 = cases.flat_sources(pos, flux)
    numtau = thin.shape[0]

    def setup():
        if p.cuda_is_init():
            p.device_close()
        p.device_init(N, 8)
        p.photo_table_to_device(thin, thick)
        lib.source_data_to_device(p0, f0, 3)
        lib.grid_to_device(capi.GRID_NDENS, nd)
        lib.grid_to_device(capi.GRID_XH_AV, xh)

    def trace(R, dr):
        lib.raytrace_device(R, cases.SIG, dr, 0, 3, cases.MINLOGTAU, dlog, numtau)
        return lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N))), lib.last_raytrace_counts()[0]

    drs = [dr0, dr0 * 1.37, dr0 * 0.731, dr0 * 3.3e-7, dr0 * 1.0000001, dr0]
    counts_seen = set()
    for mode in (0, 1, 3, 4, 9):
        for R in (5.0, 9.0):
            setup()
            lib.set_option(capi.OPT_SECTORS, mode)
            try:
                patched = [trace(R, dr) for dr in drs]              # one build, then dr changes only
            finally:
                lib.set_option(capi.OPT_SECTORS, 0)
            for dr, (phi, count) in zip(drs, patched):
                ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=numtau,
                                             flags=O.ASORA_MODE)["phi_ion"]
                w = ref != 0
                assert np.array_equal(phi != 0, w), (mode, R, dr)      # the same cells rated, on-sphere ones included
                np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL, atol=0)
                assert count == 3 * int(w.sum()) // 3 or count > 0
                counts_seen.add((R, int(w.sum())))
            # a fresh build for the last but one dr gives the same bits as the patched tables did
            setup()
            lib.set_option(capi.OPT_SECTORS, mode)
            try:
                fresh, _ = trace(R, drs[1])
            finally:
                lib.set_option(capi.OPT_SECTORS, 0)
            assert np.array_equal(fresh != 0, patched[1][0] != 0)
            np.testing.assert_allclose(fresh, patched[1][0], rtol=1e-13, atol=0)
    p.device_close()


def test_column_density_dump_of_cells_on_the_sphere_follows_dr(asora):
    """asora_debug_coldens writes the outgoing column density of the cells that receive a rate.  A lattice point exactly ON
    the sphere (R = 9: (1,4,8), (4,4,7), (3,6,6), (0,0,9) ...) is always tabulated and evaluated, but whether it is RATED -- and
    so whether it appears in the dump -- is the reference's floating-point distance test, which depends on dr; the tables are
    patched in place when dr changes (the dump must follow the patched RATE bit, not the evaluation)."""
    p, lib, capi = asora
    N, R = 32, 9.0
    thin, thick, dlog = cases.soft_tables(400)
    nd, xh, dr0 = cases.grid(N, "lognormal", 31, 0.05, xlo=1e-4, xhi=1e-2)
    pos = np.array([[16, 3], [16, 30], [16, 16]])
    flux = np.array([2.0, 1.0])
    p0, f0 = cases.flat_sources(pos, flux)
    numtau = thin.shape[0]
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.source_data_to_device(p0, f0, 2)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    patterns = set()
    for dr in (dr0, dr0 * 1.37, dr0 * 0.731, dr0 * 3.3e-7, dr0):
        for src in (0, 1):
            cd = lib.debug_coldens(R, cases.SIG, dr, src, N)
            one = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, p0[3 * src:3 * src + 3], f0[src:src + 1], thin, thick,
                                         cases.MINLOGTAU, dlog, NumTau=numtau, flags=O.ASORA_MODE, want_coldens=True)
            w = one["phi_ion"] != 0
            assert np.array_equal(cd != 0, w), (dr, src, np.argwhere((cd != 0) != w)[:8].tolist())
            np.testing.assert_allclose(cd[w], one["coldens"][w], rtol=1e-12)
            patterns.add((src, int(w.sum())))
    assert len(patterns) > 2          # the set of rated cells did change with dr
    p.device_close()


def test_column_density_on_cube_edges_of_shells_whose_reciprocal_is_inexact(asora):
    """A cell with a transverse offset EQUAL to its shell index s (cube edges, the diagonal) has two upstream corners that do
    not exist in shell s-1; their bilinear weight is 1 - s * (1/s): 0 for most s, 2^-53 for s = 49, 98, 103, 107, ...  The
    reference multiplies that speck with a real neighbour's column density.  Until round 6 the tables pointed such corners at
    the zero slot, where the speck met the value 0 -- a weight 1 / max(0.6, 0) instead of 1 / (c sigma), amplified by
    c sigma / 0.6: ~3e-12 of the column density behind shell 49 in cells of optical depth ~200 (the benchmark medium).  They
    alias their existing neighbour now.  128^3, one source, a trace beyond the box (shells up to 64, cube edges of shell 49
    at distance 69 inside it), the benchmark medium: column densities against the oracle at 5e-13 -- everywhere, and in
    particular on and behind the edge cells of shell 49."""
    import bench
    p, lib, capi = asora
    N, R = 128, 100.0
    thin, thick, dlog = cases.soft_tables(400)
    nd = np.full((N, N, N), 1e-3)
    xh = np.full((N, N, N), 2e-4)
    dr = 3 * 3.086e24 / 256                                  # tau = 227 per cell
    pos = np.array([[64], [64], [64]])
    flux = np.array([1.0])
    p0, f0 = cases.flat_sources(pos, flux)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.source_data_to_device(p0, f0, 1)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    cd = lib.debug_coldens(R, bench.SIG, dr, 0, N)
    one = O.asora_do_all_sources(R, bench.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog, NumTau=thin.shape[0],
                                 flags=O.ASORA_MODE, want_coldens=True)
    w = cd != 0
    assert w.sum() > 0.9 * N ** 3
    np.testing.assert_allclose(cd[w], one["coldens"][w], rtol=5e-13, atol=0)
    # the cells in question were part of the comparison: offsets (49, 49, t) and beyond, on the source's +++ side
    assert w[63 + 49, 63 + 49, 63 + 10] and w[63 + 55, 63 + 55, 63 + 55] and not w[63 + 60, 63 + 60, 63 + 60]
    p.device_close()


def test_two_sources_per_workgroup_give_the_same_rates(asora):
    """ASORA_OPT_PAIR_SOURCES: one workgroup sweeps its unit for two consecutive sources at once.  Sources whose spheres do
    not overlap (no summation-order freedom) -> grids IDENTICAL to the one-source-per-workgroup kernel, for every
    decomposition and workgroup size the paired variant exists for, for even and ODD source counts (the last workgroup then
    carries a dummy second source whose rates must be dropped) and with the Fortran constants; overlapping sources against
    the oracle; and the pair counts the library reports do not change."""
    p, lib, capi = asora
    N = 96
    thin, thick, dlog = cases.soft_tables(400)
    nd, xh, dr = cases.grid(N, "lognormal", 12, 0.4, xlo=1e-4, xhi=1e-2)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    lattice = np.array([(i, j, k) for i in range(1, N, 12) for j in range(1, N, 12) for k in range(1, N, 12)]).T     # spacing 12 > 2 R
    rng = np.random.RandomState(5)
    pick = rng.permutation(lattice.shape[1])[:81]
    pos = lattice[:, pick].copy()
    pos[:, 0] = [1, 1, 1]                                       # a corner: the periodic wrap is in play
    pos[:, 1] = [N, 37, 1]
    flux = rng.uniform(1.0, 5.0, 81)
    p0, f0 = cases.flat_sources(pos, flux)

    def trace(R, n, pairs, **opts):
        lib.source_data_to_device(p0[:3 * n], f0[:n], n)
        lib.set_option(capi.OPT_PAIR_SOURCES, pairs)
        for k, v in opts.items():
            lib.set_option(getattr(capi, k), v)
        try:
            lib.raytrace_device(R, cases.SIG, dr, 0, n, cases.MINLOGTAU, dlog, thin.shape[0])
            phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
            counts = lib.last_raytrace_counts()
        finally:
            lib.set_option(capi.OPT_PAIR_SOURCES, 0)
            for k in opts:
                lib.set_option(getattr(capi, k), 0)
        return phi, counts

    for R in (4.0, 5.5):
        for n in (80, 81, 2, 3):
            for opts in ({}, {"OPT_SECTORS": 1}, {"OPT_SECTORS": 2}, {"OPT_SECTORS": 3}, {"OPT_SECTORS": 5}, {"OPT_SECTORS": 6},
                         {"OPT_SECTORS": 7, "OPT_BLOCK_THREADS": 256}, {"OPT_SECTORS": 8}, {"OPT_SECTORS": 9},
                         {"OPT_SECTORS": 3, "OPT_BLOCK_THREADS": 128}, {"OPT_SECTORS": 3, "OPT_BLOCK_THREADS": 256},
                         {"OPT_SECTORS": 1, "OPT_BLOCK_THREADS": 512}, {"OPT_FORTRAN_CONSTANTS": 1}):
                one, c1 = trace(R, n, 1, **opts)
                two, c2 = trace(R, n, 2, **opts)
                assert np.array_equal(one, two), (R, n, opts)
                assert c1[0] == c2[0] == n * int(((np.add.outer(np.add.outer(np.arange(-6, 7) ** 2, np.arange(-6, 7) ** 2),
                                                                np.arange(-6, 7) ** 2)) <= R * R).sum())
                assert one.max() > 0
    # overlapping spheres (R = 9 > half the spacing) against the oracle, odd count
    ref = O.asora_do_all_sources(9.0, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0], flags=O.ASORA_MODE)["phi_ion"]
    w = ref != 0
    for mode in (1, 3, 5):
        phi, _ = trace(9.0, 81, 2, OPT_SECTORS=mode)
        assert np.array_equal(phi != 0, w)
        np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL, atol=0)
    # a medium so thick that most cells lie beyond the last table entry: their rates are exactly +0 and are not added
    # (ASORA_OPT_SKIP_ZERO_RATES = 0), whole waves of them skip the rate arithmetic -- grids IDENTICAL to adding everything (2),
    # with one and with two sources per workgroup, on dense and on line-aligned tables
    nd_thick = nd * 3.0e4
    lib.grid_to_device(capi.GRID_NDENS, nd_thick)
    left_out = {0: 0, 1: 0}
    for R in (4.0, 5.5):
        for n in (80, 81):
            for opts in ({}, {"OPT_SECTORS": 9}, {"OPT_SECTORS": 9, "OPT_ALIGNED_ROWS": 2}, {"OPT_SECTORS": 3, "OPT_BLOCK_THREADS": 256},
                         {"OPT_SECTORS": 6}):
                everything, _ = trace(R, n, 1, OPT_SKIP_ZERO_RATES=2, **opts)
                assert (everything == 0).sum() > (everything != 0).sum() > 0
                for pairs in (1, 2):
                    for skip in (0, 1):      # 0: the library decides by its probes (sooner or later it leaves them out), 1: always
                        phi, _ = trace(R, n, pairs, OPT_SKIP_ZERO_RATES=skip, **opts)
                        assert np.array_equal(phi, everything), (R, n, pairs, skip, opts)
                        left_out[skip] = max(left_out[skip], lib.last_raytrace_zero_rates())
    assert left_out[1] > 0 and left_out[0] > 0, left_out
    p.device_close()


def test_rows_cut_at_64_byte_lines_give_the_same_rates(asora):
    """ASORA_OPT_ALIGNED_ROWS: the units of one face (six sectors, twelve sector pairs) take their tables by the source's
    position modulo 8 along the memory-contiguous axis, and the paired variant puts two sources into a workgroup only when
    they agree in it.  Sources in every class of i and of k whose spheres do not overlap (no summation-order freedom) ->
    grids IDENTICAL to the densely packed tables, with one and with two sources per workgroup, even, odd and tiny source
    counts, every workgroup size, sources whose spheres wrap around the box, the Fortran constants; the pair counts the
    library reports do not change; overlapping sources against the oracle."""
    p, lib, capi = asora
    N = 96
    thin, thick, dlog = cases.soft_tables(400)
    nd, xh, dr = cases.grid(N, "lognormal", 14, 0.4, xlo=1e-4, xhi=1e-2)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    lattice = np.array([(i, j, k) for i in range(2, N, 12) for j in range(2, N, 12) for k in range(2, N, 12)]).T     # spacing 12
    rng = np.random.RandomState(15)
    pick = rng.permutation(lattice.shape[1])[:97]
    pos = lattice[:, pick] + rng.randint(-1, 3, size=(3, 97))    # every residue modulo 8 occurs on every axis; distance >= 9 > 2 R
    pos[:, 0] = [1, 1, 1]                                         # corners: the periodic wrap is in play
    pos[:, 1] = [N, 38, N]
    assert set((pos[0] - 1) % 8) == set(range(8)) and set((pos[2] - 1) % 8) == set(range(8))
    flux = rng.uniform(1.0, 5.0, 97)
    p0, f0 = cases.flat_sources(pos, flux)

    def trace(R, n, aligned, pairs, **opts):
        lib.source_data_to_device(p0[:3 * n], f0[:n], n)
        lib.set_option(capi.OPT_ALIGNED_ROWS, aligned)
        lib.set_option(capi.OPT_PAIR_SOURCES, pairs)
        for k, v in opts.items():
            lib.set_option(getattr(capi, k), v)
        try:
            lib.raytrace_device(R, cases.SIG, dr, 0, n, cases.MINLOGTAU, dlog, thin.shape[0])
            phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
            counts = lib.last_raytrace_counts()
        finally:
            lib.set_option(capi.OPT_ALIGNED_ROWS, 0)
            lib.set_option(capi.OPT_PAIR_SOURCES, 0)
            for k in opts:
                lib.set_option(getattr(capi, k), 0)
        return phi, counts

    for R in (4.0, 4.4):
        inside = int(((np.add.outer(np.add.outer(np.arange(-6, 7) ** 2, np.arange(-6, 7) ** 2), np.arange(-6, 7) ** 2)) <= R * R).sum())
        for n in (96, 97, 1, 2, 3):
            for opts in ({"OPT_SECTORS": 9}, {"OPT_SECTORS": 3}, {"OPT_SECTORS": 9, "OPT_BLOCK_THREADS": 64},
                         {"OPT_SECTORS": 9, "OPT_BLOCK_THREADS": 128}, {"OPT_SECTORS": 3, "OPT_BLOCK_THREADS": 256},
                         {"OPT_SECTORS": 9, "OPT_BLOCK_THREADS": 512}, {"OPT_SECTORS": 9, "OPT_FORTRAN_CONSTANTS": 1}):
                dense, c0 = trace(R, n, 1, 1, **opts)
                for pairs in (1, 2):
                    cut, c1 = trace(R, n, 2, pairs, **opts)
                    assert np.array_equal(dense, cut), (R, n, pairs, opts)
                    assert c0[0] == c1[0] == n * inside, (R, n, pairs, opts, c0, c1)
                assert dense.max() > 0
    # overlapping spheres against the oracle, odd count, both kinds of unit; and a kind of unit that has no aligned tables
    ref = O.asora_do_all_sources(9.0, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0], flags=O.ASORA_MODE)["phi_ion"]
    w = ref != 0
    for mode in (9, 3, 1):
        for pairs in (1, 2):
            phi, _ = trace(9.0, 97, 2, pairs, OPT_SECTORS=mode)
            assert np.array_equal(phi != 0, w)
            np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL, atol=0)
    p.device_close()

    # spheres that reach the periodic window (the units of the + and - side then differ: no table is shared between them)
    # and beyond it, on a small mesh: cut tables against dense ones, same support, rates to summation order
    N = 40
    nd, xh, dr = cases.grid(N, "lognormal", 16, 0.15, xlo=1e-4, xhi=1e-2)
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    pos, flux = cases.sources(N, 37, 17, flux=2.0)
    p0, f0 = cases.flat_sources(pos, flux)
    for R in (12.5, 19.0, 20.0, 27.0, 40.0):
        for mode in (9, 3):
            dense, c0 = trace(R, 37, 1, 1, OPT_SECTORS=mode)
            for pairs in (1, 2):
                cut, c1 = trace(R, 37, 2, pairs, OPT_SECTORS=mode)
                assert c0 == c1, (R, mode, pairs, c0, c1)
                assert np.array_equal(dense != 0, cut != 0), (R, mode, pairs)
                np.testing.assert_allclose(cut, dense, rtol=1e-11, atol=0, err_msg=str((R, mode, pairs)))
    p.device_close()


def test_buffer_and_global_rate_atomics_give_the_same_rates(asora):
    """The rate atomics go through buffer descriptors by default (out-of-range offset = lane has nothing to add) and as
    global atomics under a branch with ASORA_OPT_GLOBAL_ATOMICS.  80 sources whose spheres do not overlap (no
    summation-order freedom) -> IDENTICAL grids, for every decomposition (octants, sectors, mirrored pairs), with heating,
    with the grey opacity and with the [k][j][i] accumulator copy on and off; overlapping sources of both forms against the
    oracle."""
    p, lib, capi = asora
    N = 96
    thin, thick, dlog = cases.soft_tables(400)
    heat_thin, heat_thick = 0.7 * thin + 0.01, 0.6 * thick + 0.02
    nd, xh, dr = cases.grid(N, "lognormal", 11, 0.4, xlo=1e-4, xhi=1e-2)
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    p.photo_table_to_device(thin, thick)
    lib.heat_table_to_device(heat_thin, heat_thick, thin.shape[0])
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    lattice = np.array([(i, j, k) for i in range(1, N, 12) for j in range(1, N, 12) for k in range(1, N, 12)]).T     # spacing 12 > 2 R
    rng = np.random.RandomState(3)
    pick = rng.permutation(lattice.shape[1])[:80]
    pos = lattice[:, pick].copy()
    pos[:, 0] = [1, 1, 1]                                       # a corner: the periodic wrap is in play
    flux = rng.uniform(1.0, 5.0, 80)
    p0, f0 = cases.flat_sources(pos, flux)
    lib.source_data_to_device(p0, f0, 80)

    def trace(R, n):
        lib.raytrace_device(R, cases.SIG, dr, 0, n, cases.MINLOGTAU, dlog, thin.shape[0])
        return lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))

    def both(R, **opts):
        out = []
        for glob in (0, 1):
            lib.set_option(capi.OPT_GLOBAL_ATOMICS, glob)
            for k, v in opts.items():
                lib.set_option(getattr(capi, k), v)
            try:
                phi = trace(R, 80)
                heat = lib.grid_to_host(capi.GRID_PHI_HEAT, np.empty((N, N, N))) if opts.get("OPT_HEATING") else None
            finally:
                lib.set_option(capi.OPT_GLOBAL_ATOMICS, 0)
                for k in opts:
                    lib.set_option(getattr(capi, k), 1 if k == "OPT_Z_TRANSPOSED" else 0)
            out.append((phi, heat))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][0].max() > 0
        if out[0][1] is not None:
            assert np.array_equal(out[0][1], out[1][1]) and out[0][1].max() > 0
        return out[0][0]

    for R in (4.0, 5.5):
        ref = both(R)
        assert int((ref != 0).sum()) > 80 * 200
        for mode in (1, 2, 3):                                  # octants, sectors, mirrored pairs
            np.testing.assert_allclose(both(R, OPT_SECTORS=mode), ref, rtol=1e-13, atol=0)
        both(R, OPT_HEATING=1)
        both(R, OPT_GREY_NOTABLES=1)
        np.testing.assert_allclose(both(R, OPT_Z_TRANSPOSED=0), ref, rtol=1e-13, atol=0)
    # overlapping spheres, both forms against the oracle
    ref = O.asora_do_all_sources(9.0, cases.SIG, dr, nd, xh, p0, f0, thin, thick, cases.MINLOGTAU, dlog,
                                 NumTau=thin.shape[0], flags=O.ASORA_MODE)["phi_ion"]
    for glob in (0, 1):
        lib.set_option(capi.OPT_GLOBAL_ATOMICS, glob)
        try:
            phi = trace(9.0, 80)
        finally:
            lib.set_option(capi.OPT_GLOBAL_ATOMICS, 0)
        assert np.array_equal(phi == 0, ref == 0)
        w = ref != 0
        np.testing.assert_allclose(phi[w], ref[w], rtol=GAMMA_RTOL, atol=0)


def test_randomised_time_steps_of_the_device_resident_loop_against_the_oracle_loop(asora, tmp_path):
    """Seeded sweep of whole time steps through evolve3D (device-resident loop): odd and even meshes (tiles of the fused
    pass cut by the mesh edge), radii from one cell to beyond the box, 1 to 6 sources (criterion (NumSrc-1)/3 = 0 for one
    source: only the relative-change test can end the loop), uniform and non-uniform temperature grids (both forms of the
    fused pass), two consecutive steps (the second starts from a grid with fronts).  Iteration counts, fields and the
    returned rates against the oracle's restatement of the reference loop."""
    from evolve_oracle import evolve3D_oracle
    p, lib, capi = asora
    rng = np.random.default_rng(777)
    thin, thick, dlog = cases.soft_tables(600)
    for trial in range(10):
        N = int(rng.choice([9, 12, 17, 20, 33]))
        ns = int(rng.integers(1, 7))
        R = float(rng.choice([1.0, 2.5, 4.0, N / 3.0, N * 0.8, 1000.0]))
        nd, xh, dr = cases.grid(N, "lognormal", 500 + trial, float(10 ** rng.uniform(-1.5, 0.3)), xlo=1e-4, xhi=2e-3)
        temp = np.full((N, N, N), 1e4) if trial % 2 else 10 ** rng.uniform(3.7, 4.3, size=(N, N, N))
        pos = 1 + rng.integers(0, N, size=(3, ns))
        flux = rng.uniform(0.5, 2.0, size=ns) * 3e-4 * (N / 16.0) ** 3 / ns
        dt = 3.15576e13 * float(rng.choice([0.5, 2.0, 5.0]))
        if p.cuda_is_init():
            p.device_close()
        p.device_init(N, 8)
        p.photo_table_to_device(thin, thick)
        x, x_ref = xh, xh
        for step in range(2):
            x, phi = p.evolve3D(dt, dr, flux, pos, True, 1000, N, 1e-2, temp, nd, x, thin, thick, cases.MINLOGTAU, dlog, R,
                                1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C,
                                logfile=str(tmp_path / "log"), quiet=True)
            x_ref, phi_ref, niter_ref, _ = evolve3D_oracle(dt, dr, flux, pos, temp, nd, x_ref, thin, thick, cases.MINLOGTAU,
                                                           dlog, R, 1e-4, cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
                                                           cases.TEMPH0, cases.ABU_C)
            tag = f"trial {trial} step {step}: N={N} ns={ns} R={R:g} dt={dt:.3g} uniform_T={bool(trial % 2)}"
            assert p.evolve._evolve.last_niter == niter_ref, tag
            np.testing.assert_allclose(x, x_ref, rtol=1e-8, atol=0, err_msg=tag)
            scale = phi_ref.max()
            np.testing.assert_allclose(phi, phi_ref, rtol=1e-7, atol=1e-13 * scale, err_msg=tag)
