"""First call (host builds and uploads the geometry tables of the radius) against second call of a whole-box trace, and
what a change of dr ALONE costs at an integer radius with lattice points on the sphere (a cosmological run changes dr
every time step; the cells on the sphere are re-classified in place, nothing is rebuilt).
usage: python tools/time_geometry_build.py [--N 128 256 320]   -- prints one JSON line per case"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pyc2ray_amd as p
from pyc2ray_amd import _capi
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources
ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, nargs="+", default=[128, 256, 320])
ap.add_argument("--nsrc", type=int, nargs="+", default=[1, 1000])
a = ap.parse_args()
lib = load_asora()
thin, thick, dlog = bench.make_tables()
for N in a.N:
    for ns in a.nsrc:
        if p.cuda_is_init():
            p.device_close()
        p.device_init(N, 64)
        p.photo_table_to_device(thin, thick)
        ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
        p0, f0 = format_sources(pos, flux)
        lib.source_data_to_device(p0, f0, ns)
        lib.grid_to_device(_capi.GRID_NDENS, ndens)
        lib.grid_to_device(_capi.GRID_XH_AV, xh)
        lib.synchronize()
        R = 0.9 * N          # beyond the box: every cell of the periodic window is reached
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0] - 1)
            lib.synchronize()
            times.append(time.perf_counter() - t0)
        print(json.dumps({"N": N, "sources": ns, "R": R, "first_call_s": times[0], "second_call_s": times[1], "third_call_s": times[2],
                          "geometry_build_and_upload_s": times[0] - times[1], "geometry_table_MB": lib.debug_geometry_bytes() / 1e6,
                          "variant": lib.last_raytrace_variant()}), flush=True)
# dr-only change: 256^3, 1000 sources, integer radii (lattice points exactly on the sphere: 6 at R = 64, 30 at R = 25, ...)
for R in (64.0, 30.0, 25.0):
    N, ns = 256, 1000
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 64)
    p.photo_table_to_device(thin, thick)
    ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
    p0, f0 = format_sources(pos, flux)
    lib.source_data_to_device(p0, f0, ns)
    lib.grid_to_device(_capi.GRID_NDENS, ndens)
    lib.grid_to_device(_capi.GRID_XH_AV, xh)

    def call(dr_now):
        lib.synchronize()
        t0 = time.perf_counter()
        lib.raytrace_device(R, bench.SIG, dr_now, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0] - 1)
        lib.synchronize()
        return time.perf_counter() - t0

    first = call(dr)
    same = min(call(dr) for _ in range(5))
    changed = [call(dr * (1.0 + 0.013 * (q + 1))) for q in range(5)]
    print(json.dumps({"N": N, "sources": ns, "R": R, "first_call_s": first, "steady_call_ms": same * 1e3,
                      "call_after_dr_change_ms": [c * 1e3 for c in changed],
                      "dr_change_costs_ms": (min(changed) - same) * 1e3}), flush=True)
p.device_close()
