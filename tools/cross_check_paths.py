"""Consistency of the two GPU raytracers at a size the CPU checkers cannot reach: the ASORA path (sphere of radius R,
Fortran-flavoured constants switched on) against the sub-box path (cube, rates within R) on the same inputs.
usage: python tools/cross_check_paths.py [--N 512] [--nsrc 300] [--R 32] [--workload cosmo]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=512)
ap.add_argument("--nsrc", type=int, default=300)
ap.add_argument("--R", type=float, default=32.0)
ap.add_argument("--workload", default="cosmo")
a = ap.parse_args()
N, ns, R = a.N, a.nsrc, a.R
lib = load_asora()
thin, thick, dlog = bench.make_tables()
ndens, xh, temp, dr, pos, flux = bench.make_workload(a.workload, N, ns)
flux = np.full(ns, float(flux.mean()))          # equal fluxes: the sub-box path rates every source with the last one's
lib.device_init_auto(N)
lib.photo_table_to_device(thin, thick, thin.shape[0])
p0, f0 = format_sources(pos, flux)
lib.source_data_to_device(p0, f0, ns)
lib.grid_to_device(0, ndens)
lib.grid_to_device(1, xh)
lib.set_option(0, 1)                            # ASORA_OPT_FORTRAN_CONSTANTS
t0 = time.perf_counter()
lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0])
lib.synchronize()
t_a = time.perf_counter() - t0
lib.set_option(0, 0)
phi_a = lib.grid_to_host(2, np.empty((N, N, N)))
t0 = time.perf_counter()
nbox, loss = lib.subbox_raytrace_device(int(R), int(R), 0.0, R, bench.SIG, dr, bench.MINLOGTAU, dlog, thin.shape[0], 0, ns)
lib.synchronize()
t_s = time.perf_counter() - t0
phi_s = lib.grid_to_host(2, np.empty((N, N, N)))
w = phi_a != 0
rel = np.abs(phi_a[w] - phi_s[w]) / phi_a[w]
worst = int(np.argmax(rel))
print(json.dumps({"worst_cell_asora": float(phi_a[w][worst]), "worst_cell_subbox": float(phi_s[w][worst]),
                  "phi_max": float(phi_a.max()), "rel_diff_quantiles": [float(q) for q in np.quantile(rel, [0.5, 0.99, 0.9999])],"N": N, "sources": ns, "R": R, "workload": a.workload, "cells_compared": int(w.sum()),
                  "same_support": bool(np.array_equal(w, phi_s != 0)),
                  "max_rel_diff": float(np.max(np.abs(phi_a[w] - phi_s[w]) / phi_a[w])),
                  "asora_path_s": t_a, "subbox_path_s": t_s, "nsubbox": nbox}))
