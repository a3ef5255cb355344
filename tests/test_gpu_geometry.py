"""The geometry tables of the raytrace kernel: built on the device (csrc/geometry_device.hip, the default since round 5) against
the host-side builder (csrc/geometry.hip, ASORA_OPT_GEOMETRY_ON_HOST = 1), BIT FOR BIT -- every entry of every table of every
launch shape the library picks or can be forced into: whole spheres, half spheres, octants and their pairs, sectors of every
kind (3, 6, 12, 24 per source), quarter sectors (96), line-aligned forms (8 classes), the sub-box tables with their triple
padding, clipped periodic windows (radius beyond the box, even and odd meshes), radii with lattice points exactly on the sphere.
Equal tables mean equal results: every parity test of the suite then holds for both builders; one case checks the rates anyway,
and the in-place re-classification of the on-sphere cells when dr changes.  (The reference derives its geometry inside the
kernel, src/asora/raytracing.cu:39-59,228-238: no counterpart to compare with but the results.)"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asora():
    import pyc2ray_amd as p
    from pyc2ray_amd import _capi
    from pyc2ray_amd.load_extensions import load_asora
    yield p, load_asora(), _capi
    if p.cuda_is_init():
        p.device_close()


def _setup(p, lib, capi, N, ns, seed=3):
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 8)
    thin, thick, dlog = cases.soft_tables()
    p.photo_table_to_device(thin, thick)
    nd, xh, dr = cases.grid(N, "lognormal", seed, 0.1)
    rng = np.random.default_rng(seed)
    pos = 1 + rng.integers(0, N, size=(3, ns))
    flux = rng.uniform(0.5, 2.0, size=ns)
    p0, f0 = cases.flat_sources(pos, flux)
    lib.source_data_to_device(p0, f0, ns)
    lib.grid_to_device(capi.GRID_NDENS, nd)
    lib.grid_to_device(capi.GRID_XH_AV, xh)
    return thin, dlog, dr


def _tables(lib, capi, on_host, call):
    lib.set_option(capi.OPT_GEOMETRY_ON_HOST, 1 if on_host else 0)
    call()
    t, meta = lib.debug_geometry_tables()
    meta["variant"] = {k: v for k, v in lib.last_raytrace_variant().items() if k != "skip_zero"}     # (that one follows the probes of the medium)
    return t, meta


def _same(dev, host, what):
    (td, md), (th, mh) = dev, host
    assert md == mh, (what, md, mh)
    assert len(td) == len(th)
    for q, (a, b) in enumerate(zip(td, th)):
        assert a.shape == b.shape, (what, q, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.flatnonzero((a != b).any(axis=1))
            raise AssertionError(f"{what}: table {q} differs in {bad.size} of {a.shape[0]} entries, first at {bad[0]}: "
                                 f"device {a[bad[0]]} host {b[bad[0]]}")


# (N, sources, R, {option: value}): the launch shapes of pick_launch_shape and the forced ones
SHAPES = [
    (64, 700, 8.0, {}), (64, 700, 13.0, {}), (96, 1000, 18.0, {}), (96, 1000, 22.5, {}), (96, 1000, 24.5, {}),
    (128, 1000, 30.0, {}),                      # six sectors, line-aligned, paired (integer radius: lattice points on the sphere)
    (128, 1000, 32.5, {}), (128, 1000, 41.0, {}),       # six sectors / twelve pairs, aligned
    (128, 1000, 56.0, {}),                      # twelve pairs x 512, packed
    (128, 300, 60.0, {"OPT_ALIGNED_ROWS": 2}),  # aligned on request beyond the default range
    (64, 100, 20.0, {}), (64, 30, 25.0, {}),    # fewer workgroups than the chip wants: the round-2 table
    (64, 20, 20.0, {}),                         # 24 sectors
    (64, 1, 20.0, {}), (48, 2, 1000.0, {}),     # 96 quarter sectors; whole box, clipped window
    (64, 1000, 1000.0, {}), (33, 1000, 1000.0, {}), (40, 4, 15.0, {}),          # whole box: even and odd mesh
    (64, 20, 1000.0, {}), (64, 3, 1000.0, {"OPT_SECTORS": 4, "OPT_BLOCK_THREADS": 256}), (64, 1000, 40.0, {}),   # clipped windows: tables that share their inner shells
    (64, 400, 14.0, {"OPT_SECTORS": 1, "OPT_BLOCK_THREADS": 64}), (64, 400, 14.0, {"OPT_SECTORS": 5, "OPT_BLOCK_THREADS": 128}),
    (64, 400, 17.0, {"OPT_SECTORS": 8}), (64, 400, 17.0, {"OPT_SECTORS": 9, "OPT_BLOCK_THREADS": 128}),
    (64, 400, 17.0, {"OPT_SECTORS": 2, "OPT_BLOCK_THREADS": 64}), (64, 400, 17.0, {"OPT_SECTORS": 7, "OPT_BLOCK_THREADS": 512}),
    (64, 400, 26.0, {"OPT_SECTORS": 3, "OPT_BLOCK_THREADS": 1024}), (64, 3, 30.0, {"OPT_SECTORS": 4, "OPT_BLOCK_THREADS": 256}),
    (32, 50, 0.5, {}), (32, 50, 1.0, {}), (32, 50, 1.8, {}),
    # a sphere that touches the window of a tiny mesh (r = N/2): families of small tables that share their inner shells, one after the
    # other in one process -- every forced shape, aligned ones among them
    *[(16, 1, 8.0, {"OPT_SECTORS": m, "OPT_BLOCK_THREADS": 256}) for m in (1, 2, 3, 4, 6, 9)],
]


@pytest.mark.parametrize("N,ns,R,opts", SHAPES)
def test_device_built_tables_equal_host_built_tables(asora, N, ns, R, opts):
    p, lib, capi = asora
    thin, dlog, dr = _setup(p, lib, capi, N, ns)
    for k, v in opts.items():
        lib.set_option(getattr(capi, k), v)
    call = lambda: lib.raytrace_device(R, cases.SIG, dr, 0, ns, cases.MINLOGTAU, dlog, thin.shape[0] - 1)
    try:
        dev = _tables(lib, capi, False, call)
        phi_dev = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        host = _tables(lib, capi, True, call)
        phi_host = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
    finally:
        lib.set_option(capi.OPT_GEOMETRY_ON_HOST, 0)
        for k in opts:
            lib.set_option(getattr(capi, k), 0)
    _same(dev, host, (N, ns, R, opts))
    assert np.array_equal(phi_dev != 0, phi_host != 0)
    np.testing.assert_allclose(phi_dev, phi_host, rtol=1e-11, atol=0)       # (the same tables; the atomics' order is free)


@pytest.mark.parametrize("N,ns,R,box,tables,pair", [(64, 600, 20.0, 5, 2, 0), (64, 600, 9.0, 3, 2, 0), (96, 800, 30.0, 7, 2, 2),
                                                    (64, 600, 16.0, 100, 2, 1), (128, 1000, 32.0, 8, 0, 0)])
def test_device_built_sub_box_tables_equal_host_built_ones(asora, N, ns, R, box, tables, pair):
    """The tables of the sub-box sweep (raytracing.f90:127-249 on tabulated geometry): the traversal range instead of the periodic
    window, no octahedron bound, whole triples of steps behind every sub-box boundary."""
    p, lib, capi = asora
    thin, dlog, dr = _setup(p, lib, capi, N, ns, seed=5)
    lib.set_option(capi.OPT_SUBBOX_TABLES, tables)
    lib.set_option(capi.OPT_PAIR_SOURCES, pair)
    res = {}
    call = lambda: res.__setitem__("r", lib.subbox_raytrace_device(N, box, 0.0, R, cases.SIG, dr, cases.MINLOGTAU, dlog, thin.shape[0] - 1, 0, ns))
    try:
        dev = _tables(lib, capi, False, call)
        r_dev = res["r"]
        host = _tables(lib, capi, True, call)
        r_host = res["r"]
    finally:
        lib.set_option(capi.OPT_GEOMETRY_ON_HOST, 0)
        lib.set_option(capi.OPT_SUBBOX_TABLES, 0)
        lib.set_option(capi.OPT_PAIR_SOURCES, 0)
    _same(dev, host, (N, ns, R, box))
    assert r_dev[0] == r_host[0] and abs(r_dev[1] - r_host[1]) <= 1e-10 * abs(r_host[1])


def test_cells_on_the_sphere_follow_dr_with_both_builders(asora):
    """Integer radius with lattice points on the sphere ((6, 8, 0) at R = 10, ...): whether such a cell is rated is the reference's
    floating-point distance test, which depends on dr (raytracing.cu:302-305,315); the tables are patched in place when dr
    changes.  The device-built tables must carry the same patch list: same tables after the same sequence of dr."""
    p, lib, capi = asora
    N, ns, R = 64, 500, 10.0
    thin, dlog, dr = _setup(p, lib, capi, N, ns, seed=9)
    out = {}
    for on_host in (False, True):
        lib.set_option(capi.OPT_GEOMETRY_ON_HOST, 1 if on_host else 0)
        seq = []
        for f in (1.0, 1.0 + 2.0 ** -40, 0.7310585786300049, 1.0):
            lib.raytrace_device(R, cases.SIG, dr * f, 0, ns, cases.MINLOGTAU, dlog, thin.shape[0] - 1)
            t, meta = lib.debug_geometry_tables()
            seq.append((t, meta, lib.last_raytrace_counts()[0]))
        out[on_host] = seq
    lib.set_option(capi.OPT_GEOMETRY_ON_HOST, 0)
    for q, (a, b) in enumerate(zip(out[False], out[True])):
        _same((a[0], a[1]), (b[0], b[1]), ("dr sequence", q))
        assert a[2] == b[2]



@pytest.mark.parametrize("ns,limit_MB", [(1000, 30.0), (1, 60.0)])
def test_whole_box_tables_share_their_inner_shells(asora, ns, limit_MB):
    """A trace beyond the box on an even mesh: the periodic window is [-N/2, N/2 - 1], every sign variant of a unit needs a table of
    its own, and they differ in the last shell only.  The variants' tables share the memory of everything before it (the first
    one holds it, the kernel reads the others' inner steps through it; geometry_device.hip): at 128^3 the twelve sector-pair tables of a many-source
    trace take 21 MB instead of 70, the 96 quarter-sector tables of a single source 52 MB instead of 160 -- with the contents (checked
    bit for bit against the host builder above) and the results (here: against the oracle) unchanged."""
    from oracle import oracle as O
    p, lib, capi = asora
    N = 128
    thin, dlog, dr = _setup(p, lib, capi, N, ns, seed=11)
    R = 1000.0
    lib.raytrace_device(R, cases.SIG, dr, 0, ns, cases.MINLOGTAU, dlog, thin.shape[0] - 1)
    mb = lib.debug_geometry_bytes() / 1e6
    v = lib.last_raytrace_variant()
    assert v["units"] == (96 if ns == 1 else 12), v
    assert 5.0 < mb < limit_MB, (mb, v)
    if ns == 1:           # one source: the oracle is affordable
        nd, xh, _ = cases.grid(N, "lognormal", 11, 0.1)
        rng = np.random.default_rng(11)
        pos = 1 + rng.integers(0, N, size=(3, ns))
        flux = rng.uniform(0.5, 2.0, size=ns)
        p0, f0 = cases.flat_sources(pos, flux)
        phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N, N, N)))
        ref = O.asora_do_all_sources(R, cases.SIG, dr, nd, xh, p0, f0, thin, cases.soft_tables()[1], cases.MINLOGTAU, dlog,
                                     NumTau=thin.shape[0] - 1, flags=O.ASORA_MODE)["phi_ion"]
        w = ref != 0
        assert np.array_equal(phi != 0, w) and int(w.sum()) == N ** 3
        np.testing.assert_allclose(phi[w], ref[w], rtol=1e-8, atol=0)

