"""What ONE rank of an N-GPU run computes per outer iteration, timed on one GPU, plus the bytes it exchanges.

There is one GPU on the build box, so the multi-GPU step cannot be measured; its compute half can.  For P = 1, 2, 4, 8
ranks and BASELINE configs[3] (256^3 log-normal density, 1000 sources IN TOTAL on the densest cells, r_RT = 32) this runs,
for every rank r of the plan in turn, exactly the library calls one iteration of TorchComm.slab_enqueue makes around the exchanges
(the sharded device loop, round 5) -- the trace of r's sources (in chunks), the out-box folds of the planes that leave, the adds of
what would arrive (same bytes, from a grid of the device), the fused pass on own[r], nHI on the halo planes, the convergence
test -- and reports the slowest rank's time next to the plan's bytes per rank.  The exchange time is then MODELLED from the
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
ap.add_argument("--chunks", type=int, default=4, help="trace chunks of the overlapped slab exchange (TorchComm.slab_chunks)")
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
        K = plan.common_chunks(a.chunks)
        sched, rsched = plan.send_schedule(r, K), plan.recv_schedule(r, K)
        cb = plan.chunk_bounds(hi - lo, K)
        own_a, own_b = plan.own[r]
        halo = [(x, y) for q in range(P) if q != r for x, y in plan.runs[r][q]]          # planes whose new xh_av arrives
        lib.evolve_begin_slab(*chem, a.R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, 0, hi - lo, -1.0, 0.0, own_a, own_b - own_a)
        stand_in = lib.device_ptr(_capi.GRID_NDENS)          # "received" planes: any device memory of the right size

        def iteration(marks=None, t0=None):
            for c in range(K):
                lib.evolve_slab_trace(cb[c], cb[c + 1] - cb[c])
                for _, x, y in sched[c]:
                    lib.evolve_slab_fold_out(x, y - x)
                if marks is not None:
                    lib.synchronize()
                    marks.append(time.perf_counter() - t0)
            for c in range(K):
                for _, x, y in rsched[c]:
                    lib.evolve_slab_add(x, y - x, stand_in + 8 * N * N * x)
            lib.evolve_slab_pass()
            for x, y in halo:
                lib.evolve_slab_nhi(x, y - x)
            lib.evolve_slab_close(None)           # (the rank's own sums stand in for the all-reduced ones: nothing waits on the host)

        for _ in range(2):
            iteration()
        lib.evolve_poll(0)
        # the GPU's time per iteration: batches of 8 enqueued without a host synchronisation in between, as evolve3D_MPI does;
        # split by the library's HIP-event timers (raytrace | out-box folds + adds | fused pass | halo nHI)
        best = None
        for _ in range(a.reps):
            lib.set_option(_capi.OPT_TIMING, 1)
            lib.kernel_time_reset()
            lib.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                iteration()
            lib.synchronize()
            wall = (time.perf_counter() - t0) / 8
            rt = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)[0] / 8
            fin = lib.kernel_time_ms(_capi.KERNEL_FINISH)[0] / 8
            ch = lib.kernel_time_ms(_capi.KERNEL_CHEMISTRY)[0] / 8
            pr = lib.kernel_time_ms(_capi.KERNEL_PREP)[0] / 8
            lib.set_option(_capi.OPT_TIMING, 0)
            lib.evolve_poll(0)
            cur = (wall, (rt + fin) * 1e-3, (ch + pr) * 1e-3)
            best = cur if best is None or cur[0] < best[0] else best
        # when do the pieces of the rate exchange become available?  the same loop with a synchronisation after every chunk
        ready = None
        for _ in range(a.reps):
            marks = []
            iteration(marks, time.perf_counter())
            lib.synchronize()
            lib.evolve_poll(0)
            ready = marks if ready is None or marks[-1] < ready[-1] else ready
        # bytes per chunk on the busiest outgoing link of this rank
        plane = 8 * N * N
        per_peer = {}
        for c in range(K):
            for q, x, y in sched[c]:
                per_peer.setdefault(q, [0] * K)[c] += (y - x) * plane
        busiest = max(per_peer.values(), key=sum) if per_peer else [0] * K
        per_rank.append({"rank": r, "sources": hi - lo, "planes_worked_on": int(sum(c for _, c in work)), "own_planes": plan.own[r][1] - plan.own[r][0],
                         "compute_ms": best[0] * 1e3, "prepare_trace_fold_ms": best[1] * 1e3, "slab_chemistry_ms": best[2] * 1e3,       # (wall per iteration of a batch of 8 | kernels: trace + out-box folds + adds | fused pass + halo nHI)
                         "bytes_sent_per_exchange": plan.bytes_per_rank(r)[0], "bytes_received_per_exchange": plan.bytes_per_rank(r)[1],
                         "chunk_ready_ms": [m * 1e3 for m in ready], "busiest_link_bytes_per_chunk": busiest})
    slow = max(per_rank, key=lambda q: q["compute_ms"])
    peers = max(sum(1 for q in range(P) if q != r and (plan.runs[r][q] or plan.runs[q][r])) for r in range(P)) if P > 1 else 0
    worst_bytes = max(max(q["bytes_sent_per_exchange"], q["bytes_received_per_exchange"]) for q in per_rank)
    # a rank's transfers to different peers use different links; the exchange lasts as long as its largest single transfer
    largest_pair = plan.largest_transfer()
    exch_ms = largest_pair / (a.link_GBs * 1e9) * 1e3
    # the overlapped schedule: a link carries a rank's pieces one after the other, each from the moment its chunk is done;
    # the chemistry of a slab starts when the slowest rank's last piece has arrived; the xh_av exchange is not overlapped
    def overlapped_step_ms(link_GBs):
        arrive = 0.0
        for q in per_rank:
            t = 0.0
            for c, nbytes in enumerate(q["busiest_link_bytes_per_chunk"]):
                if nbytes:
                    t = max(t, q["chunk_ready_ms"][c]) + nbytes / (link_GBs * 1e9) * 1e3
            arrive = max(arrive, t, q["prepare_trace_fold_ms"])
        return arrive + max(q["slab_chemistry_ms"] for q in per_rank) + largest_pair / (link_GBs * 1e9) * 1e3
    rows.append({"ranks": P, "slowest_rank_compute_ms": slow["compute_ms"], "of_which_prepare_trace_fold_ms": slow["prepare_trace_fold_ms"],
                 "trace_chunks": plan.common_chunks(a.chunks),
                 "modelled_step_ms_overlapped": {f"{g:g} GB/s": (overlapped_step_ms(g) if P > 1 else slow["compute_ms"]) for g in (50.0, 100.0, 150.0)},
                 "of_which_slab_chemistry_ms": slow["slab_chemistry_ms"], "max_bytes_one_direction_per_exchange": worst_bytes,
                 "largest_single_transfer_bytes": largest_pair, "peers_of_the_busiest_rank": peers,
                 "modelled_exchange_ms_each_of_two": exch_ms, "modelled_step_ms": slow["compute_ms"] + 2 * exch_ms, "per_rank": per_rank})
t1 = rows[0]["modelled_step_ms"]
for r in rows:
    r["modelled_speedup_over_one_rank"] = t1 / r["modelled_step_ms"]
    r["modelled_speedup_overlapped"] = {k: rows[0]["slowest_rank_compute_ms"] / v for k, v in r["modelled_step_ms_overlapped"].items()}
print(json.dumps({"workload": f"{a.workload} {N}^3, {a.nsrc} sources in total, r_RT={a.R:g}",
                  "measured": "per-rank compute (library calls between the exchanges), one GPU, best of %d" % a.reps,
                  "modelled": f"exchange time = largest single rank-to-rank transfer / {a.link_GBs:g} GB/s (assumed), two exchanges per iteration; "
                              "the scalar all-gather and launch gaps are not included",
                  "rows": rows}))
p.device_close()
