"""Wall-clock time of whole time steps through the C2Ray_Test class with and without `device_resident` (BASELINE
configs[3]-like: 256^3 log-normal density, 1000 sources on the densest cells, r_RT = 32, 1 Myr steps).
Prints one JSON line.  usage: python tools/time_steps_resident.py [--N 256] [--steps 6]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pyc2ray_amd as pc2r

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--nsrc", type=int, default=1000)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
N = a.N
ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, a.nsrc)
out = {"case": f"{N}^3 log-normal density, {a.nsrc} sources, r_RT = 32, {a.steps} steps of 1 Myr through C2Ray_Test.evolve3D"}
os.chdir(tempfile.mkdtemp())
for resident in (False, True):
    if pc2r.cuda_is_init():
        pc2r.device_close()
    sim = pc2r.C2Ray_Test(os.path.join(ROOT, "tests", "data", "parameters_test.yml"), N, True)
    sim.R_max_LLS = 32.0
    sim.dr = dr
    sim.ndens = np.asfortranarray(ndens)
    sim.device_resident = resident
    sim.evolve3D(bench.MYR, flux * 30.0, pos)                       # warm-up step (geometry tables, first touch)
    t0 = time.perf_counter()
    iters = 0
    for _ in range(a.steps):
        sim.evolve3D(bench.MYR, flux * 30.0, pos)
        iters += pc2r.evolve._evolve.last_niter
    pc2r.load_extensions.load_asora().synchronize()
    t = time.perf_counter() - t0
    mean_x = float(sim.xh.mean())                                    # (resident: the first read since the run began)
    out["device_resident" if resident else "default"] = {"ms_per_time_step": t / a.steps * 1e3, "outer_iterations": iters,
                                                         "mean_x_after": mean_x}
out["speedup"] = out["default"]["ms_per_time_step"] / out["device_resident"]["ms_per_time_step"]
print(json.dumps(out))          # (the log lines of C2Ray_Test go to its logfile and to stdout: keep the LAST line)
pc2r.device_close()
