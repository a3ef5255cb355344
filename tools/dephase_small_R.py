#!/usr/bin/env python3
"""One structural attempt at the r_RT = 16 point of BASELINE configs[2]: 1000 sources are ONE round of 500 paired
workgroups on the chip's 512 slots, all in phase (profiles/r04_sweep_counts_R16.txt: 0.201 ms for 1000 sources against 0.159 ms
per 1000 at 8000).  De-phasing without a new kernel: the same sources as TWO (or four) launches on the library's two side streams
(asora_raytrace_begin + _range per part + _fold), so that the workgroups of the second launch arrive while those of the first
are already some shells into their sweep and the two sets co-reside on every CU out of phase.

Measured: wall clock of the whole call (zeroing, nHI, trace, fold) between two device synchronisations, one launch
(asora_raytrace_device) against 2 and 4 parts, interleaved, `--reps` times each; the parts see the same launch shape (the shape is
chosen for the whole list).  usage (GPU box): python tools/dephase_small_R.py [--R 16] [--nsrc 1000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=float, default=16.0)
    ap.add_argument("--nsrc", type=int, default=1000)
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import pyc2ray_amd as p
    from pyc2ray_amd import _capi
    from pyc2ray_amd.load_extensions import load_asora
    from pyc2ray_amd.utils.sourceutils import format_sources
    lib = load_asora()
    N, R, ns = args.N, args.R, args.nsrc
    p.device_init(N, 64)
    thin, thick, dlog = bench.make_tables()
    p.photo_table_to_device(thin, thick)
    numtau = thin.shape[0] - 1
    ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
    order = np.lexsort((pos[2], pos[1], pos[0]))          # neighbours in the list are neighbours in space, for every variant alike
    pos, flux = pos[:, order], flux[order]
    p0, f0 = format_sources(pos, flux)
    lib.source_data_to_device(p0, f0, ns)
    lib.grid_to_device(_capi.GRID_NDENS, ndens)
    lib.grid_to_device(_capi.GRID_XH_AV, xh)

    def one_launch():
        lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, numtau)

    def parts(k):
        def run():
            lib.raytrace_begin(R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau)
            b = [q * ns // k for q in range(k + 1)]
            for q in range(k):
                lib.raytrace_range(b[q], b[q + 1] - b[q])
            lib.raytrace_fold(0, N)
        return run

    variants = {"one_launch": one_launch, "two_parts_two_streams": parts(2), "four_parts_two_streams": parts(4)}
    ref = None
    for name, fn in variants.items():          # same sums whatever the cut
        fn()
        phi = lib.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
        if ref is None:
            ref = phi
        else:
            np.testing.assert_allclose(phi, ref, rtol=1e-11, atol=0)
    times = {k: [] for k in variants}
    for _ in range(args.reps):
        for name, fn in variants.items():
            lib.synchronize()
            t0 = time.perf_counter()
            fn()
            lib.synchronize()
            times[name].append((time.perf_counter() - t0) * 1e3)
    out = {"workload": f"{N}^3 uniform, {ns} sources, r_RT = {R:g}: the whole raytrace call (zero + nHI + trace + fold), ms",
           "variant_of_the_one_launch": lib.last_raytrace_variant(), "reps": args.reps}
    for name, v in times.items():
        v = np.sort(np.array(v))
        out[name] = {"median_ms": float(np.median(v)), "min_ms": float(v[0]), "p10_ms": float(v[len(v) // 10]), "p90_ms": float(v[-len(v) // 10])}
    print(json.dumps(out))
    p.device_close()


if __name__ == "__main__":
    main()
