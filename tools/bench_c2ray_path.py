"""The reference's raytracing benchmark, CPU leg (raytracing_benchmark/run_test.py:88: do_all_sources(normflux,
srcpos, max_subbox=1000, subboxsize=r_RT, ..., loss_fraction, tables, R_max_LLS=r_RT)), through this build's
libc2ray-compatible entry point on the GPU, and -- when oracle/_ref is present -- through the reference Fortran
itself on one host core, on the same inputs.  Prints one JSON line per radius.
usage: python tools/bench_c2ray_path.py [--N 256] [--nsrc 1000] [--R 16 32] [--cpu-sources 1000] [--workload uniform]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyc2ray_amd.load_extensions import load_asora, load_c2ray

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--nsrc", type=int, default=1000)
ap.add_argument("--R", type=float, nargs="+", default=[16.0, 32.0])
ap.add_argument("--cpu-sources", type=int, default=1000)
ap.add_argument("--workload", default="uniform")
ap.add_argument("--loss-fraction", type=float, default=1e-2)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--global-shells", type=int, default=0, help="1: ASORA_OPT_SUBBOX_GLOBAL_SHELLS (shell buffers in global memory)")
ap.add_argument("--tables", type=int, default=0, help="ASORA_OPT_SUBBOX_TABLES: 0 auto, 1 on-the-fly geometry only (round 2), 2 tabulated whenever possible")
ap.add_argument("--pair-sources", type=int, default=0, help="ASORA_OPT_PAIR_SOURCES for the tabulated sweep: 0 auto, 1 never, 2 always")
a = ap.parse_args()

N, ns = a.N, a.nsrc
c2ray = load_c2ray()
asora = load_asora()
thin, thick, dlog = bench.make_tables()
ndens, xh, temp, dr, pos, flux = bench.make_workload(a.workload, N, ns)
nd_f, xh_f = np.asfortranarray(ndens), np.asfortranarray(xh)
zeros = np.zeros(thin.shape[0])

asora.set_option(9, a.global_shells)      # ASORA_OPT_SUBBOX_GLOBAL_SHELLS
asora.set_option(14, a.tables)            # ASORA_OPT_SUBBOX_TABLES
asora.set_option(13, a.pair_sources)      # ASORA_OPT_PAIR_SOURCES
for R in a.R:
    sub = int(R)
    phi = np.zeros((N, N, N), order="F")
    heat = np.zeros((N, N, N), order="F")
    cd = np.zeros((N, N, N), order="F")
    call = lambda: c2ray.raytracing.do_all_sources(flux, pos, 1000, sub, cd, bench.SIG, dr, nd_f, xh_f, phi, heat,
                                                   a.loss_fraction, thin, thick, zeros, zeros, bench.MINLOGTAU, dlog, R)
    call()
    asora.set_option(2, 1)           # ASORA_OPT_TIMING
    asora.kernel_time_reset()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        nbox, loss = call()
    t_gpu = (time.perf_counter() - t0) / a.reps
    k_ms, k_n = asora.kernel_time_ms(0)
    asora.set_option(2, 0)
    m = int(R)
    rr = np.arange(-m, m + 1)
    rated = int(((rr[:, None, None] ** 2 + rr[None, :, None] ** 2 + rr[None, None, :] ** 2) <= R * R).sum()) * ns
    # one box of +-subboxsize per source; only the LAST source's column densities go back to the caller, so only its cube is
    # evaluated beyond the radius (subbox.hip: `beyond`)
    swept = rated + ((2 * sub + 1) ** 3 - rated // ns) if nbox == ns else None
    algo_bytes = 32 * rated + 8 * ((swept - rated) if swept else 0)                          # 32 B per rated cell, 8 B (nHI) per carried-on cell
    out = {"call": "libc2ray.raytracing.do_all_sources on the GPU (host grids in/out, Fortran order)", "N": N,
           "roofline_sweep_kernel": {"bound": "hbm", "algorithmic_bytes_per_call": algo_bytes, "rated_cells": rated, "swept_cells": swept,
                                     "achieved_GBs": algo_bytes / (k_ms / a.reps * 1e-3) / 1e9, "peak_GBs": 8000.0,
                                     "frac": algo_bytes / (k_ms / a.reps * 1e-3) / 1e9 / 8000.0}, "shell_buffers": "global memory" if a.global_shells else "LDS when they fit",
           "geometry": {0: "auto", 1: "on the fly (subbox.hip)", 2: "tabulated (raytrace.hip SUBBOX) + on the fly for the dumped source"}[a.tables],
           "pair_sources_option": a.pair_sources, "sources": ns, "R": R, "subboxsize": sub, "loss_fraction": a.loss_fraction, "s_per_call": t_gpu,
           "sweep_kernels_ms_per_call": k_ms / a.reps, "sweep_launches_per_call": k_n / a.reps,
           "nsubbox": nbox, "photon_loss": loss}
    try:
        from oracle import ref_fortran as F
        have_ref = F.available()
    except Exception:
        have_ref = False
    m = min(a.cpu_sources, ns)
    if have_ref and m > 0:
        # the reference rates every source with the LAST source's flux (raytracing.f90:500,503): keep the last
        # source last when timing a subset
        sel = np.r_[0:m - 1, ns - 1] if m < ns else np.arange(ns)
        t0 = time.perf_counter()
        r = F.do_all_sources(flux[sel], pos[:, sel], max_subbox=1000, subboxsize=sub, sig=bench.SIG, dr=dr, ndens=ndens,
                             xh_av=xh, loss_fraction=a.loss_fraction, thin=thin, thick=thick, minlogtau=bench.MINLOGTAU,
                             dlogtau=dlog, R_max_LLS=R)
        t_cpu = time.perf_counter() - t0
        out.update({"cpu_reference_s": t_cpu, "cpu_sources": int(m), "cpu_cores": 1,
                    "cpu_s_per_source": t_cpu / m, "speedup_same_sources": (t_cpu / m) * ns / t_gpu,
                    "cpu_nsubbox": r["nsubbox"], "cpu_photon_loss": r["photon_loss"]})
        if m == ns:
            ref = r["phi_ion"]
            w = ref != 0
            out["phi_max_rel_diff_vs_reference"] = float(np.max(np.abs(phi[w] - ref[w]) / ref[w]))
            out["phi_cells_compared"] = int(w.sum())
            out["same_support"] = bool(np.array_equal(w, phi != 0))
            # the ASORA path (sphere of radius R only, device-resident) on the same inputs, Fortran-flavoured
            # constants, against the same reference output
            from pyc2ray_amd.utils.sourceutils import format_sources
            p0, f0 = format_sources(pos, flux)
            asora.photo_table_to_device(thin, thick, thin.shape[0])
            asora.source_data_to_device(p0, f0, ns)
            asora.grid_to_device(0, ndens)
            asora.grid_to_device(1, xh)
            asora.set_option(0, 1)       # ASORA_OPT_FORTRAN_CONSTANTS
            asora.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0])
            asora.set_option(0, 0)
            phi_a = asora.grid_to_host(2, np.empty((N, N, N)))
            out["asora_path_max_rel_diff_vs_reference"] = float(np.max(np.abs(phi_a[w] - ref[w]) / ref[w]))
            out["asora_path_same_support"] = bool(np.array_equal(w, phi_a != 0))
    print(json.dumps(out), flush=True)
