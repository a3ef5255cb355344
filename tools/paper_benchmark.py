"""The reference's published raytracing benchmark protocol on the MI355X.

ref: test/paper_tests/raytracing_benchmark/run_test.py:22-110 -- N = 250 (not a power of two), uniform ndens = 1e-3,
xh = 2e-4, box 3 Mpc, black-body Teff = 1e5 K table with NumTau = 20000, R in {10, 30, 50, 100} cells, the first N_s
sources of a halo list for N_s = 1 ... 1e6, and per (R, N_s) the mean wall-clock time of 10 direct calls of
`libasora.do_all_sources(r_RT, coldensh_out, sig, dr, ndens, xh_av, phi_ion, nsrc, N, minlogtau, dlogtau, NumTau)`
-- each call INCLUDING the upload of xh_av and the download of phi_ion (src/asora/raytracing.cu:117,146).  The
published figure is the asymptote 3 t / (N_s 4 pi R^3) = 3.156 ns per source per in-sphere cell at R = 30,
N_s = 1e6 on a Tesla P100 (plot_sources.ipynb).

The halo list is an absent blob of the reference checkout (.MISSING_LARGE_BLOBS); the sources here are the
RandomState(100) positions of the reference's own generate_test_sourcefile (sourceutils.py:56-58), equal fluxes --
the cost of a call does not depend on positions or fluxes.  For the largest radii N_s is capped so that a call stays
below ~1 s.  Prints one JSON line.  usage: python tools/paper_benchmark.py [--reps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pyc2ray_amd as p
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--N", type=int, default=250)
ap.add_argument("--R", type=int, nargs="+", default=[10, 30, 50, 100])
ap.add_argument("--max-pairs", type=float, default=1.5e11, help="cap on N_s * (4 pi R^3 / 3) per call")
a = ap.parse_args()

N = a.N
lib = load_asora()
p.device_init(N, 64)                                    # source_batch_size 64 (parameters.yml:139); unused by this build
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
numtau = thin.shape[0] - 1                              # sim.NumTau, run_test.py:85
dr = 3 * 3.086e24 / N
ndens_flat = np.full(N ** 3, 1e-3)
xh_av_flat = np.full(N ** 3, 2e-4)
phi_ion_flat = np.zeros(N ** 3)
coldensh_out_flat = np.zeros(N ** 3)
lib.density_to_device(ndens_flat, N)

rng = np.random.RandomState(100)
all_pos = (1 + rng.randint(0, N, size=3 * 10 ** 6)).reshape((10 ** 6, 3), order="C").T.copy()
rows = []
for R in a.R:
    insphere = 4.0 * np.pi * R ** 3 / 3.0
    for nsrc in (1, 10, 100, 1000, 10 ** 4, 10 ** 5, 10 ** 6):
        if nsrc * insphere > a.max_pairs:
            break
        p0, f0 = format_sources(all_pos[:, :nsrc], np.ones(nsrc))
        lib.source_data_to_device(p0, f0, nsrc)
        call = lambda: lib.do_all_sources(float(R), coldensh_out_flat, bench.SIG, dr, ndens_flat, xh_av_flat, phi_ion_flat,
                                          nsrc, N, bench.MINLOGTAU, dlog, numtau)
        call()                                          # first call of a radius builds its geometry tables
        t0 = time.time()
        for _ in range(a.reps):
            call()
        t = (time.time() - t0) / a.reps
        rows.append({"R": R, "numsrc": nsrc, "seconds_per_call": t,
                     "ns_per_source_per_insphere_cell": t * 1e9 / (nsrc * insphere)})
        print(f"R={R} nsrc={nsrc}: {t * 1e3:.3f} ms per call, {rows[-1]['ns_per_source_per_insphere_cell']:.5f} ns", file=sys.stderr)
best = min(r["ns_per_source_per_insphere_cell"] for r in rows if r["R"] == 30) if any(r["R"] == 30 for r in rows) else None
print(json.dumps({
    "protocol": "ref test/paper_tests/raytracing_benchmark/run_test.py: N=250, uniform medium, direct calls of "
                "libasora.do_all_sources incl. the PCIe copies of xh_av and phi_ion, mean of %d calls" % a.reps,
    "published_P100_ns_per_source_per_insphere_cell_R30_Ns1e6": 3.156,
    "this_build_asymptote_R30_ns": best,
    "speedup_vs_published_asymptote": (3.156 / best) if best else None,
    "rows": rows}))
p.device_close()
