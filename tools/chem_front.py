"""The chemistry pass where it is NOT at its best case: a grid with ionisation fronts.

bench.py's medium is quiet (Gamma ~ 0 almost everywhere, every cell leaves do_chemistry after its minimum number of
iterations).  Here BASELINE configs[3] (256^3 log-normal density, 1000 sources on the densest cells) is driven with
fluxes x FLUX_SCALE, so that after one converged time step ~20 % of the volume is ionised and the next step starts with
fronts everywhere.  Reported for the quiet and for the front state, per outer iteration of ONE time step:
  * duration of the fused chemistry pass (HIP events, ASORA_OPT_TIMING) and of the raytrace,
  * the count of non-converged cells,
and for the FIRST iteration of the step (the one with the most chemistry work) the histogram of do_chemistry trip
counts per cell (src/c2ray/chemistry.f90:146-203), recomputed on the host from the downloaded inputs of that pass.
Prints one JSON line.  usage: python tools/chem_front.py [--flux-scale 1e3] [--N 256]
Under rocprofv3 --pmc (tools/pmc_chem.sh) the same run gives SQ_INSTS_VALU / SQ_WAVE_CYCLES of the chemistry kernel.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pyc2ray_amd as p
from pyc2ray_amd import _capi
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

ap = argparse.ArgumentParser()
ap.add_argument("--flux-scale", type=float, default=1e3)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--nsrc", type=int, default=1000)
ap.add_argument("--R", type=float, default=32.0)
ap.add_argument("--histogram", type=int, default=1)
a = ap.parse_args()
N = a.N
CHEM = (bench.MYR, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)


trip_counts = bench.trip_counts       # do_chemistry's trip count per cell, vectorised on the host


def one_time_step(lib, dr, dlog, numtau, nsrc, conv_fraction=1e-4, histogram=False):
    """One time step iteration by iteration; returns per-iteration kernel times and convergence counts."""
    conv_criterion = min(int(conv_fraction * N ** 3), (nsrc - 1) / 3)
    lib.evolve_begin(*CHEM, a.R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, 0, nsrc, conv_criterion, conv_fraction)
    rows, hist = [], None
    done = False
    while not done and len(rows) < 200:
        x_before = lib.grid_to_host(_capi.GRID_XH if not rows else _capi.GRID_XH_AV, np.empty((N, N, N))) if (histogram and not rows) else None
        lib.kernel_time_reset()
        lib.evolve_enqueue(1)
        _, done, r = lib.evolve_poll(4)
        ch_ms, _ = lib.kernel_time_ms(_capi.KERNEL_CHEMISTRY)
        rt_ms, _ = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)
        rows.append({"chemistry_ms": ch_ms, "raytrace_ms": rt_ms, "nonconverged": int(r[0][0]), "rel_change": float(r[0][3])})
        if x_before is not None:
            g = lib.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
            nd = lib.grid_to_host(_capi.GRID_NDENS, np.empty((N, N, N)))
            tp = lib.grid_to_host(_capi.GRID_TEMP, np.empty((N, N, N)))
            x0 = lib.grid_to_host(_capi.GRID_XH, np.empty((N, N, N)))
            nit = trip_counts(CHEM[0], nd, tp, x0, x_before, g, *CHEM[1:])
            c = np.bincount(nit.ravel(), minlength=12)
            hist = {"trip_count_cells": {str(k): int(v) for k, v in enumerate(c) if v}, "mean_trip_count": float(nit.mean()),
                    "max_trip_count": int(nit.max()),
                    "mean_of_wave_maxima": float(nit.reshape(-1, 64).max(axis=1).mean())}
    return rows, hist


def main():
    lib = load_asora()
    p.device_init(N, 64)
    thin, thick, dlog = bench.make_tables()
    p.photo_table_to_device(thin, thick)
    numtau = thin.shape[0] - 1
    ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, a.nsrc)
    out = {"workload": f"BASELINE configs[3] ({N}^3 log-normal, {a.nsrc} sources on the densest cells, r_RT={a.R:g}), dt = 1 Myr",
           "flux_scale_of_the_front_state": a.flux_scale}
    lib.set_option(_capi.OPT_TIMING, 1)
    for label, scale in (("quiet", 1.0), ("fronts", a.flux_scale)):
        p0, f0 = format_sources(pos, flux * scale)
        lib.source_data_to_device(p0, f0, a.nsrc)
        lib.grid_to_device(_capi.GRID_NDENS, ndens)
        lib.grid_to_device(_capi.GRID_TEMP, temp)
        lib.grid_to_device(_capi.GRID_XH, xh)
        rows1, _ = one_time_step(lib, dr, dlog, numtau, a.nsrc)                  # step 1: from the neutral grid
        x1 = lib.grid_to_host(_capi.GRID_XH_INTERMED, np.empty((N, N, N)))
        lib.grid_to_device(_capi.GRID_XH, x1)
        rows2, hist = one_time_step(lib, dr, dlog, numtau, a.nsrc, histogram=bool(a.histogram))   # step 2: starts with the fronts of step 1
        out[label] = {
            "ionised_volume_fraction_after_step_1": float((x1 > 0.5).mean()), "mean_x_after_step_1": float(x1.mean()),
            "step_1": {"outer_iterations": len(rows1), "chemistry_ms": [round(r["chemistry_ms"], 4) for r in rows1],
                       "nonconverged": [r["nonconverged"] for r in rows1]},
            "step_2": {"outer_iterations": len(rows2), "chemistry_ms": [round(r["chemistry_ms"], 4) for r in rows2],
                       "raytrace_ms": [round(r["raytrace_ms"], 4) for r in rows2],
                       "nonconverged": [r["nonconverged"] for r in rows2], "first_iteration": hist},
        }
    q = np.mean(out["quiet"]["step_2"]["chemistry_ms"])
    f = max(out["fronts"]["step_1"]["chemistry_ms"] + out["fronts"]["step_2"]["chemistry_ms"])
    out["slowest_front_pass_over_quiet_pass"] = f / q
    p.device_close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
