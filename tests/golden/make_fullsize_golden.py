"""Sparse golden vectors at the FULL benchmark size, from the REFERENCE ITSELF.

Run in the build container (where /root/reference exists), one radius per process (each takes minutes):

    make -C oracle
    python tests/golden/make_fullsize_golden.py 16 &
    python tests/golden/make_fullsize_golden.py 32 &
    python tests/golden/make_fullsize_golden.py 64 &

BASELINE.json configs[2]: 256^3 uniform medium, the 1000 RandomState(100) sources of bench.py, the
Teff = 1e5 K black-body table with NumTau = 20000, r_RT in {16, 32, 64}.  The expected rates come from
oracle/_ref/libc2ray_ref.so -- the reference's own src/c2ray/*.f90 compiled with flang -- swept over one
sub-box of +-r_RT cells per source with R_max_LLS = r_RT (ref: test/paper_tests/raytracing_benchmark/
run_test.py:88), which deposits rates on exactly the cells the ASORA path rates (|d| <= r_RT).

A dense 256^3 grid is 128 MiB; what is stored per radius is a few hundred KB:
  vals         Gamma at 30 000 seeded cell indices (regenerated from the seed at test time)
  src_vals     Gamma at the 1000 source cells (the largest values of the grid)
  plane_sums   sum of Gamma over every i-plane (256 values)
  block_sums   sum over every 16^3 block (4096 values): with plane_sums a checksum over ALL cells
  nonzero      number of cells with Gamma != 0, total = sum of Gamma
  table_sums   sums of the thin / thick tables the run used (the test regenerates them with the same code)
Data only: inputs are regenerated from seeds, nothing of the reference's source text is kept.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))

import bench  # noqa: E402  (workload + table builders shared with the benchmark and the test)
from oracle import ref_fortran as F  # noqa: E402

N, NS, NSAMPLE = 256, 1000, 30000


def sample_indices(R):
    """Seeded flat C-order cell indices the fixture holds values for."""
    return np.random.default_rng(20260000 + int(R)).choice(N ** 3, size=NSAMPLE, replace=False)


def digest(phi):
    """phi: (N,N,N) logical [i,j,k] -> the checksums stored in / compared with the fixture."""
    phi = np.ascontiguousarray(phi)
    B = N // 16
    return dict(plane_sums=phi.sum(axis=(1, 2)),
                block_sums=phi.reshape(B, 16, B, 16, B, 16).sum(axis=(1, 3, 5)).ravel(),
                nonzero=np.array(int(np.count_nonzero(phi))), total=np.array(float(phi.sum())))


def main(R):
    assert F.available(), "build oracle/_ref first: make -C oracle"
    thin, thick, dlog = bench.make_tables()
    ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, NS)
    t0 = time.time()
    r = F.do_all_sources(flux, pos, max_subbox=int(R), subboxsize=int(R), sig=bench.SIG, dr=dr, ndens=ndens,
                         xh_av=xh, loss_fraction=0.0, thin=thin, thick=thick, minlogtau=bench.MINLOGTAU,
                         dlogtau=dlog, R_max_LLS=float(R), NumTau=thin.shape[0] - 1)
    phi = np.ascontiguousarray(r["phi_ion"])
    print(f"R={R}: reference took {time.time() - t0:.0f} s, nsubbox={r['nsubbox']}", flush=True)
    flat = phi.ravel()
    src_flat = ((pos[0] - 1) * N + (pos[1] - 1)) * N + (pos[2] - 1)
    out = digest(phi)
    out.update(vals=flat[sample_indices(R)], src_vals=flat[src_flat], table_sums=np.array([thin.sum(), thick.sum()]),
               nsubbox=np.array(r["nsubbox"]))
    np.savez_compressed(os.path.join(HERE, f"fullsize_uniform_R{int(R)}.npz"), **out)
    print(f"R={R}: nonzero={int(out['nonzero'])}, total={float(out['total']):.6e}", flush=True)


if __name__ == "__main__":
    main(float(sys.argv[1]))
