"""First call (host builds and uploads the geometry tables of the radius) against second call of a whole-box trace.
usage: python tools/time_geometry_build.py [--N 128 256 320]   -- prints one JSON line per mesh"""
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
                          "geometry_build_and_upload_s": times[0] - times[1]}), flush=True)
p.device_close()
