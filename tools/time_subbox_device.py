"""The sub-box sweep (libc2ray.raytracing.do_all_sources semantics) on DEVICE-RESIDENT inputs, no column-density dump: every
source goes through the tabulated kernel (raytrace.hip SUBBOX), so the event timer is that kernel alone.  One JSON line per
(radius, pair option, table form).  usage (GPU box): python tools/time_subbox_device.py [--R 16 32] [--pairs 1 2] [--aligned 1 2] [--reps 10]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pyc2ray_amd as p
from pyc2ray_amd import _capi
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--nsrc", type=int, default=1000)
ap.add_argument("--R", type=float, nargs="+", default=[16.0, 32.0])
ap.add_argument("--pairs", type=int, nargs="+", default=[1, 2])
ap.add_argument("--aligned", type=int, nargs="+", default=[1, 2], help="ASORA_OPT_ALIGNED_ROWS: 1 = packed tables, 2 = line-aligned (sectors only: r >= 25.5), 0 = the library's choice")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--loss-fraction", type=float, default=1e-2)
a = ap.parse_args()
N, ns = a.N, a.nsrc
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
p0, f0 = format_sources(pos, flux)
lib.source_data_to_device(p0, f0, ns)
lib.grid_to_device(_capi.GRID_NDENS, ndens)
lib.grid_to_device(_capi.GRID_XH_AV, xh)
for R in a.R:
    for rnd in (1, 2):
      for aligned in a.aligned:
        for pairs in a.pairs:
            lib.set_option(_capi.OPT_PAIR_SOURCES, pairs)
            lib.set_option(_capi.OPT_ALIGNED_ROWS, aligned)
            call = lambda: lib.subbox_raytrace_device(1000, int(R), a.loss_fraction, R, bench.SIG, dr, bench.MINLOGTAU, dlog, thin.shape[0] - 1, 0, ns)
            call()
            lib.set_option(_capi.OPT_TIMING, 1)
            lib.kernel_time_reset()
            for _ in range(a.reps):
                nbox, loss = call()
            ms, n = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)
            lib.set_option(_capi.OPT_TIMING, 0)
            m = int(R)
            rr = np.arange(-m, m + 1)
            rated = int(((rr[:, None, None] ** 2 + rr[None, :, None] ** 2 + rr[None, None, :] ** 2) <= R * R).sum()) * ns
            print(json.dumps({"R": R, "pair_sources_option": pairs, "aligned_rows_option": aligned, "variant": lib.last_raytrace_variant(), "round": rnd, "sweep_kernel_ms_per_call": ms / a.reps, "launches_per_call": n / a.reps,
                              "nsubbox": nbox, "photon_loss": loss, "roofline_frac_hbm": 32.0 * rated / (ms / a.reps * 1e-3) / 8e12}), flush=True)
lib.set_option(_capi.OPT_PAIR_SOURCES, 0)
lib.set_option(_capi.OPT_ALIGNED_ROWS, 0)
p.device_close()
