"""When does the library build the eight-fold (line-aligned) geometry tables?  Run with ASORA_GEOM_TIMING=1: every build prints its
size to stderr.  Expected: aligned for the first radius, dense right after a change of radius, aligned again once a radius has
served 32 CALLS (note_call_radius: one per raytrace call or evolve time step; ASORA_OPT_ALIGNED_ROWS = 0).  usage (GPU box): ASORA_GEOM_TIMING=1 python tools/check_aligned_heuristic.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import pyc2ray_amd as p
from pyc2ray_amd import _capi
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources
N, ns = 256, 1000
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
p0, f0 = format_sources(pos, flux)
lib.source_data_to_device(p0, f0, ns)
lib.grid_to_device(_capi.GRID_NDENS, ndens)
lib.grid_to_device(_capi.GRID_XH_AV, xh)
def run(R, n):
    for _ in range(n):
        lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0] - 1)
    lib.synchronize()
print("first radius 30", file=sys.stderr); run(30.0, 3)
print("radius 31 (changed): 10 calls", file=sys.stderr); run(31.0, 10)
print("radius 32 (changed): 40 calls", file=sys.stderr); run(32.0, 40)
print("done", file=sys.stderr)
p.device_close()
