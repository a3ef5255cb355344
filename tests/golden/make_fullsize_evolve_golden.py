"""One whole time step of BASELINE configs[3] at its FULL size, from the REFERENCE'S kernels driven by the reference's loop.

Run in the build container (where /root/reference exists; ~5 min on 4 cores, ~6 GB):

    make -C oracle
    python tests/golden/make_fullsize_evolve_golden.py

Workload: bench.make_workload("cosmo", 256, 1000) -- 256^3 log-normal density, the 1000 densest cells as sources with
fluxes proportional to the density (x 30, so that 1 Myr ionises a few per cent of the volume and the loop needs about a dozen
iterations) -- r_RT = 32, T = 1e4 K, xh = 2e-4, dt = 1 Myr, convergence_fraction = 1e-4, the benchmark's Teff = 1e5 K table.  What runs per outer iteration is the compiled reference Fortran (oracle/_ref):
`do_all_sources` once per source with NumSrc = 1 (its multi-source form rates every source with the last source's flux,
ref: src/c2ray/raytracing.f90:500,503), one sub-box of +-32 cells, R_max_LLS = 32, loss_fraction = 0 -- the cells the ASORA
path rates -- summed in source order; then `global_pass` (ref: src/c2ray/chemistry.f90:13).  The loop around them is the one
of ref: pyc2ray/evolve.py:127-240, restated here line by line (initial copies :136-137, conv_criterion :127, the two sums
:216-217, relative changes :219-227, the test :232, the previous sums :234-235); the reference's own evolve3D cannot be
used for this case because of the flux quirk above (its CPU branch hands all sources to ONE do_all_sources call).

Stored (sparse, ~0.5 MB): number of outer iterations, per iteration (conv_flag, sum x, sum 1-x), and of the final
ionised fraction and the last iteration's rates: 30 000 seeded cells within the sources' spheres, the source cells, per-plane
and per-16^3-block sums (make_bigconfig_golden.digest), plus checksums of the regenerated inputs.  Outputs only.
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
sys.path.insert(0, HERE)

import bench  # noqa: E402
import make_bigconfig_golden as MB  # noqa: E402
from oracle import ref_fortran as F  # noqa: E402

N, NS, R = 256, 1000, 32
DT, CONV_FRACTION = bench.MYR, 1e-4
FLUX_SCALE = 30.0          # the workload's fluxes (mean 1e48 photons/s) x 30: a partially ionised box after 1 Myr, ~a dozen outer iterations
_G = {}


def _trace(job):
    lo, hi = job
    nd_f, xh_av_f, pos, flux, thin, thick, dlog, dr = (_G[k] for k in ("nd_f", "xh_av_f", "pos", "flux", "thin", "thick", "dlog", "dr"))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    r = lambda x: C.byref(C.c_double(x))
    i = lambda x: C.byref(C.c_int(x))
    phi = np.zeros((N, N, N), order="F")
    heat = np.zeros((N, N, N), order="F")
    cd = np.zeros((N, N, N), order="F")
    total = np.zeros((N, N, N))
    zeros = np.zeros(thin.shape[0])
    lib = F.lib()
    rng1 = np.arange(-R, R + 1)
    for s in range(lo, hi):
        p1 = np.asfortranarray(pos[:, s:s + 1].astype(np.int32))
        f1 = np.array([flux[s]], dtype=np.float64)
        nbox, loss = C.c_int(0), C.c_double(0.0)
        lib._QMraytracingPdo_all_sources(
            d(f1), p1.ctypes.data_as(C.POINTER(C.c_int32)), i(R), i(R), d(cd), r(bench.SIG), r(dr), d(nd_f), d(xh_av_f), d(phi),
            d(heat), C.byref(nbox), C.byref(loss), C.byref(C.c_float(0.0)), d(thin), d(thick), d(zeros), d(zeros),
            r(bench.MINLOGTAU), r(dlog), r(float(R)), i(thin.shape[0] - 1), i(1), i(N), i(N), i(N))
        sel = np.ix_(*(((int(pos[a, s]) - 1 + rng1) % N) for a in range(3)))
        total[sel] += phi[sel]
    return total


def main():
    assert F.available(), "build oracle/_ref first: make -C oracle"
    thin, thick, dlog = bench.make_tables()
    ndens, xh, temp, dr, pos, flux0 = bench.make_workload("cosmo", N, NS)
    flux = flux0 * FLUX_SCALE
    ncell = N ** 3
    conv_criterion = min(int(CONV_FRACTION * ncell), (NS - 1) / 3)                 # evolve.py:127
    prev1 = prev0 = 2 * ncell                                                        # evolve.py:130-131
    nd_f, temp_f, xh_f = np.asfortranarray(ndens), np.asfortranarray(temp), np.asfortranarray(xh)
    xh_av, xh_intermed = xh_f.copy(order="F"), xh_f.copy(order="F")                  # evolve.py:136-137
    _G.update(nd_f=nd_f, pos=pos, flux=flux, thin=thin, thick=thick, dlog=dlog, dr=dr)
    rows, converged, niter, phi = [], False, 0, None
    workers = 4
    jobs = [(w * NS // workers, (w + 1) * NS // workers) for w in range(workers)]
    t00 = time.time()
    while not converged:
        niter += 1
        t0 = time.time()
        _G["xh_av_f"] = xh_av
        with mp.get_context("fork").Pool(workers) as pool:           # (forked per iteration: the children see this iteration's xh_av)
            parts = pool.map(_trace, jobs)
        phi = parts[0]
        for part in parts[1:]:
            phi += part
        xh_av, xh_intermed, conv_flag = F.global_pass(DT, nd_f, temp_f, xh_f, xh_av, xh_intermed, np.asfortranarray(phi), bench.BH00,
                                                      bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)
        sum1, sum0 = float(np.sum(xh_intermed)), float(np.sum(1.0 - xh_intermed))   # evolve.py:216-217
        rel1 = abs((sum1 - prev1) / sum1) if sum1 > 0.0 else 1.0                    # evolve.py:219-227
        rel0 = abs((sum0 - prev0) / sum0) if sum0 > 0.0 else 1.0
        converged = (conv_flag < conv_criterion) or (rel1 < CONV_FRACTION and rel0 < CONV_FRACTION)   # evolve.py:232
        prev1, prev0 = sum1, sum0                                                   # evolve.py:234-235
        rows.append((conv_flag, sum1, sum0, rel1, rel0))
        print(f"iteration {niter}: {time.time() - t0:.0f} s, non-converged {conv_flag}, rel change {rel1:.2e}", flush=True)
    x = np.ascontiguousarray(xh_intermed)
    out = {"niter": np.array(niter), "rows": np.array(rows), "conv_criterion": np.array(conv_criterion), "dt": np.array(DT),
           "table_sums": np.array([thin.sum(), thick.sum()])}
    out.update(MB.input_checksums(ndens, pos, flux0))
    out["flux_scale"] = np.array(FLUX_SCALE)
    idx = MB.sample_indices(N, pos, 20260400)
    src_flat = ((pos[0] - 1) * N + (pos[1] - 1)) * N + (pos[2] - 1)
    for name, grid in (("x", x), ("phi", np.ascontiguousarray(phi))):
        dg = MB.digest(grid)
        out.update({f"{name}_vals": grid.ravel()[idx], f"{name}_src_vals": grid.ravel()[src_flat], f"{name}_plane_sums": dg["plane_sums"],
                    f"{name}_block_sums": dg["block_sums"], f"{name}_total": dg["total"]})
    out["x_mean"] = np.array(float(x.mean()))
    np.savez_compressed(os.path.join(HERE, "fullsize_evolve_cosmo256.npz"), **out)
    print(f"{niter} outer iterations in {time.time() - t00:.0f} s, <x> = {x.mean():.6e}, max x = {x.max():.4f}", flush=True)


if __name__ == "__main__":
    main()
