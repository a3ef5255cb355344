"""What ONE rank of an N-GPU run computes per outer iteration, timed on one GPU, plus the bytes it exchanges.

There is one GPU on the build box, so the multi-GPU step cannot be measured; its compute half can.  For P = 1, 2, 4, 8
ranks and BASELINE configs[3] (256^3 log-normal density, 1000 sources IN TOTAL on the densest cells, r_RT = 32) this runs,
for every rank r of the plan in turn, exactly the library calls TorchComm.slab_iteration makes between the exchanges --
asora_raytrace_begin_planes on reach[r] | own[r], the trace of r's sources, the folds, asora_chemistry_range on own[r] --
and reports the slowest rank's time next to the plan's bytes per rank.  The exchange time is then MODELLED from the
bytes at an assumed point-to-point xGMI rate (printed), not measured.
Prints one JSON line.  usage: python tools/slab_compute_model.py [--workload cosmo] [--link-GBs 50]
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
from pyc2ray_amd import _capi
from pyc2ray_amd.dist import SlabPlan, TorchComm
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cosmo")
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--nsrc", type=int, default=1000)
ap.add_argument("--R", type=float, default=32.0)
ap.add_argument("--link-GBs", type=float, default=50.0, help="assumed achieved point-to-point rate per direction and peer")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
N = a.N
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
numtau = thin.shape[0] - 1
ndens, xh, temp, dr, pos, flux = bench.make_workload(a.workload, N, a.nsrc)
lib.grid_to_device(_capi.GRID_NDENS, ndens)
lib.grid_to_device(_capi.GRID_TEMP, temp)
lib.grid_to_device(_capi.GRID_XH, xh)
lib.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)
chem = (bench.MYR, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)
rows = []
for P in (1, 2, 4, 8):
    spos, sflux, bounds = TorchComm.shard_sources_by_slab(pos, flux, P)
    plan = SlabPlan(N, P, a.R, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(P)])
    per_rank = []
    for r in range(P):
        lo, hi = bounds[r], bounds[r + 1]
        p0, f0 = format_sources(spos[:, lo:hi], sflux[lo:hi])
        lib.source_data_to_device(p0, f0, hi - lo)
        work = [(x, y - x) for x, y in plan.work_runs(r)]
        lib.raytrace_begin_planes(a.R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, [(0, N)])      # a time step's first iteration
        lib.raytrace_range(0, hi - lo)
        lib.synchronize()
        best = None
        for _ in range(a.reps):
            t0 = time.perf_counter()
            lib.raytrace_begin_planes(a.R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, work)
            lib.raytrace_range(0, hi - lo)
            for x, y in plan.reach_runs(r):
                lib.raytrace_fold(x, y - x)
            lib.synchronize()
            t1 = time.perf_counter()
            x, y = plan.own[r]
            lib.chemistry_range(*chem, x, y - x, True)
            lib.chemistry_finish()
            t2 = time.perf_counter()
            cur = (t2 - t0, t1 - t0, t2 - t1)
            best = cur if best is None or cur[0] < best[0] else best
        per_rank.append({"rank": r, "sources": hi - lo, "planes_worked_on": int(sum(c for _, c in work)), "own_planes": plan.own[r][1] - plan.own[r][0],
                         "compute_ms": best[0] * 1e3, "prepare_trace_fold_ms": best[1] * 1e3, "slab_chemistry_ms": best[2] * 1e3,
                         "bytes_sent_per_exchange": plan.bytes_per_rank(r)[0], "bytes_received_per_exchange": plan.bytes_per_rank(r)[1]})
    slow = max(per_rank, key=lambda q: q["compute_ms"])
    peers = max(sum(1 for q in range(P) if q != r and (plan.run[r][q] or plan.run[q][r])) for r in range(P)) if P > 1 else 0
    worst_bytes = max(max(q["bytes_sent_per_exchange"], q["bytes_received_per_exchange"]) for q in per_rank)
    # a rank's transfers to different peers use different links; the exchange lasts as long as its largest single transfer
    largest_pair = max([(b - aa) * 8 * N * N for r in range(P) for q in range(P) if q != r and plan.run[r][q] for aa, b in [plan.run[r][q]]] or [0])
    exch_ms = largest_pair / (a.link_GBs * 1e9) * 1e3
    rows.append({"ranks": P, "slowest_rank_compute_ms": slow["compute_ms"], "of_which_prepare_trace_fold_ms": slow["prepare_trace_fold_ms"],
                 "of_which_slab_chemistry_ms": slow["slab_chemistry_ms"], "max_bytes_one_direction_per_exchange": worst_bytes,
                 "largest_single_transfer_bytes": largest_pair, "peers_of_the_busiest_rank": peers,
                 "modelled_exchange_ms_each_of_two": exch_ms, "modelled_step_ms": slow["compute_ms"] + 2 * exch_ms, "per_rank": per_rank})
t1 = rows[0]["modelled_step_ms"]
for r in rows:
    r["modelled_speedup_over_one_rank"] = t1 / r["modelled_step_ms"]
print(json.dumps({"workload": f"{a.workload} {N}^3, {a.nsrc} sources in total, r_RT={a.R:g}",
                  "measured": "per-rank compute (library calls between the exchanges), one GPU, best of %d" % a.reps,
                  "modelled": f"exchange time = largest single rank-to-rank transfer / {a.link_GBs:g} GB/s (assumed), two exchanges per iteration; "
                              "the scalar all-gather and launch gaps are not included",
                  "rows": rows}))
p.device_close()
