"""Sparse golden vectors for BASELINE.json configs[3] and configs[4] at their FULL grid sizes, from the REFERENCE ITSELF.

Run in the build container (where /root/reference exists):

    make -C oracle
    python tests/golden/make_bigconfig_golden.py cosmo256     # configs[3]: 256^3, all 1000 sources        (~1 min, 4 cores)
    python tests/golden/make_bigconfig_golden.py cosmo512     # configs[4]: 512^3, 256 of the 1e5 sources  (~3 min, 4 cores)

The workloads are bench.make_workload("cosmo", N, nsrc): log-normal density, sources on the densest cells, fluxes
proportional to the density -- UNEQUAL fluxes, so the reference Fortran is called once per source with NumSrc = 1
(its do_all_sources rates every source with normflux(NumSrc), ref: src/c2ray/raytracing.f90:500,503) and the rates are
summed here, in source order.  One sub-box of +-r_RT cells with R_max_LLS = r_RT and loss_fraction = 0 deposits rates on
exactly the cells the ASORA path rates (ref: test/paper_tests/raytracing_benchmark/run_test.py:85-88).

configs[4] names 1e5 sources; the reference on one core needs ~1 s per source at 512^3 (it zeroes two 1 GiB grids per
call), so the fixture holds the sum over a SUBSET of 256 of them (subset_indices below: spread through the list, i.e.
from the densest cell to the 1e5-th densest).  The GPU test traces exactly that subset against this fixture, and the
full list for counts, finiteness, superposition and fused-loop equality.

Stored per workload (a few hundred KB):
  vals         Gamma at 30 000 seeded cells INSIDE the spheres of the sources (regenerated from the seed at test time)
  src_vals     Gamma at the source cells
  plane_sums   sum over every i-plane;  block_sums: sum over every 16^3 block  -- together a checksum over ALL cells
  nonzero, total
  ndens_sum, pos_sum, flux_sum   checksums of the regenerated inputs (the test asserts them before comparing)
Data only: inputs are regenerated from seeds, nothing of the reference's source text is kept.
"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))

import bench  # noqa: E402
from oracle import ref_fortran as F  # noqa: E402

R = 32
NSAMPLE = 30000
WORKLOADS = {
    # name: (N, sources of the workload, sources in the fixture)
    "cosmo256": (256, 1000, 1000),
    "cosmo512": (512, 100000, 256),
}


def subset_indices(nsrc_total, nsub):
    """Indices into the source list (ordered from the densest cell down) of the sources the fixture covers."""
    if nsub >= nsrc_total:
        return np.arange(nsrc_total)
    return (np.arange(nsub) * (nsrc_total // nsub)).astype(np.int64)


def sample_indices(N, pos, seed):
    """Seeded flat C-order indices of cells within R of one of the sources `pos` ((3, ns), 1-based)."""
    rng = np.random.default_rng(seed)
    ns = pos.shape[1]
    out = np.empty(NSAMPLE, dtype=np.int64)
    n = 0
    while n < NSAMPLE:
        m = 2 * (NSAMPLE - n)
        s = rng.integers(0, ns, size=m)
        d = rng.integers(-R, R + 1, size=(3, m))
        keep = (d ** 2).sum(axis=0) <= R * R
        c = (pos[:, s[keep]] - 1 + d[:, keep]) % N
        flat = (c[0] * N + c[1]) * N + c[2]
        take = min(flat.size, NSAMPLE - n)
        out[n:n + take] = flat[:take]
        n += take
    return out


def digest(phi):
    """phi: (N,N,N) logical [i,j,k] -> the checksums stored in / compared with the fixture."""
    phi = np.ascontiguousarray(phi)
    N = phi.shape[0]
    B = N // 16
    return dict(plane_sums=phi.sum(axis=(1, 2)),
                block_sums=phi.reshape(B, 16, B, 16, B, 16).sum(axis=(1, 3, 5)).ravel(),
                nonzero=np.array(int(np.count_nonzero(phi))), total=np.array(float(phi.sum())))


def input_checksums(ndens, pos, flux):
    return dict(ndens_sum=np.array(float(ndens.sum())), pos_sum=np.array(int((pos.astype(np.int64) * [[1], [1000], [1000000]]).sum())),
                flux_sum=np.array(float(flux.sum())))


_G = {}


def _worker(job):
    """Sum of the reference's rates over the sources job = (lo, hi) of the fixture's list, as a dense C-order grid."""
    lo, hi = job
    N, nd_f, pos, flux, thin, thick, dlog, dr = (_G[k] for k in ("N", "nd_f", "pos", "flux", "thin", "thick", "dlog", "dr"))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    r = lambda x: C.byref(C.c_double(x))
    i = lambda x: C.byref(C.c_int(x))
    xh_f = np.full((N, N, N), 2e-4, order="F")
    phi = np.zeros((N, N, N), order="F")
    heat = np.zeros((N, N, N), order="F")
    cd = np.zeros((N, N, N), order="F")
    total = np.zeros((N, N, N))
    zeros = np.zeros(thin.shape[0])
    numtau = thin.shape[0] - 1                   # as raytracing_benchmark/run_test.py:85 passes it
    lib = F.lib()
    rng1 = np.arange(-R, R + 1)
    for s in range(lo, hi):
        p1 = np.asfortranarray(pos[:, s:s + 1].astype(np.int32))
        f1 = np.array([flux[s]], dtype=np.float64)
        nbox, loss = C.c_int(0), C.c_double(0.0)
        lib._QMraytracingPdo_all_sources(
            d(f1), p1.ctypes.data_as(C.POINTER(C.c_int32)), i(R), i(R), d(cd), r(bench.SIG), r(dr), d(nd_f), d(xh_f), d(phi),
            d(heat), C.byref(nbox), C.byref(loss), C.byref(C.c_float(0.0)), d(thin), d(thick), d(zeros), d(zeros),
            r(bench.MINLOGTAU), r(dlog), r(float(R)), i(numtau), i(1), i(N), i(N), i(N))
        assert nbox.value == 1
        # rates are non-zero only within R of the source: add that (periodically wrapped) cube
        ii, jj, kk = ((int(pos[a, s]) - 1 + rng1) % N for a in range(3))
        sel = np.ix_(ii, jj, kk)
        total[sel] += phi[sel]
    return total


def main(name):
    assert F.available(), "build oracle/_ref first: make -C oracle"
    N, nsrc_total, nsub = WORKLOADS[name]
    thin, thick, dlog = bench.make_tables()
    t0 = time.time()
    ndens, xh, temp, dr, pos_all, flux_all = bench.make_workload("cosmo", N, nsrc_total)
    sub = subset_indices(nsrc_total, nsub)
    pos, flux = pos_all[:, sub], flux_all[sub]
    print(f"{name}: workload built in {time.time() - t0:.0f} s; {nsub} of {nsrc_total} sources", flush=True)
    _G.update(N=N, nd_f=np.asfortranarray(ndens), pos=pos, flux=flux, thin=thin, thick=thick, dlog=dlog, dr=dr)
    workers = 4
    jobs = [(w * nsub // workers, (w + 1) * nsub // workers) for w in range(workers)]
    t0 = time.time()
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_worker, jobs)
    phi = parts[0]
    for part in parts[1:]:              # in source order
        phi += part
    print(f"{name}: reference took {time.time() - t0:.0f} s on {workers} cores", flush=True)
    flat = phi.ravel()
    src_flat = ((pos[0] - 1) * N + (pos[1] - 1)) * N + (pos[2] - 1)
    out = digest(phi)
    out.update(input_checksums(ndens, pos_all, flux_all))
    out.update(vals=flat[sample_indices(N, pos, 20260300 + N)], src_vals=flat[src_flat],
               table_sums=np.array([thin.sum(), thick.sum()]), nsub=np.array(nsub), nsrc_total=np.array(nsrc_total))
    np.savez_compressed(os.path.join(HERE, f"fullsize_{name}_R{R}.npz"), **out)
    print(f"{name}: nonzero={int(out['nonzero'])}, total={float(out['total']):.6e}, "
          f"sampled values non-zero: {int(np.count_nonzero(out['vals']))} of {NSAMPLE}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
