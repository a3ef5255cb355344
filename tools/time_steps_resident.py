"""Wall-clock time of whole time steps through the C2Ray_Test class: the class default (round 6: `device_resident`, the grids
stay on the device between steps) against `device_resident = False` (everything through the host every step, as the reference)
-- BASELINE configs[3]-like: 256^3 log-normal density, 1000 sources on the densest cells, r_RT = 32, 1 Myr steps; optionally as a
cosmological run (the density is diluted every step: on the device in the default mode).
Writes ONE JSON object to --out (the class logs to stdout).  usage: python tools/time_steps_resident.py [--N 256] [--steps 6]
[--cosmological 1] --out profiles/rNN_time_steps_resident.json"""
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
ap.add_argument("--cosmological", type=int, default=0)
ap.add_argument("--out", default=None)
a = ap.parse_args()
N = a.N
ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, a.nsrc)
out = {"case": f"{N}^3 log-normal density, {a.nsrc} sources, r_RT = 32, {a.steps} steps of 1 Myr through C2Ray_Test.evolve3D"
               + (" + cosmo_evolve (cosmological run: density diluted every step)" if a.cosmological else "")}
out_path = os.path.abspath(a.out) if a.out else None
os.chdir(tempfile.mkdtemp())
for resident in (False, None):          # None: whatever the class defaults to
    if pc2r.cuda_is_init():
        pc2r.device_close()
    sim = pc2r.C2Ray_Test(os.path.join(ROOT, "tests", "data", "parameters_test.yml"), N, True)
    sim.R_max_LLS = 32.0
    sim.dr = dr
    sim.ndens = np.asfortranarray(ndens)
    if resident is not None:
        sim.device_resident = resident
    sim.cosmological = bool(a.cosmological)
    sim.evolve3D(bench.MYR, flux * 30.0, pos)                       # warm-up step (geometry tables, first touch)
    t0 = time.perf_counter()
    iters = 0
    for _ in range(a.steps):
        if a.cosmological:
            sim.cosmo_evolve(bench.MYR)
        sim.evolve3D(bench.MYR, flux * 30.0, pos)
        iters += pc2r.evolve._evolve.last_niter
    pc2r.load_extensions.load_asora().synchronize()
    t = time.perf_counter() - t0
    mean_x = float(sim.xh.mean())                                    # (resident: the first read since the run began)
    key = "through_the_host_every_step" if resident is False else "class_default"
    out[key] = {"device_resident": bool(sim.device_resident), "ms_per_time_step": t / a.steps * 1e3, "outer_iterations": iters,
                "mean_x_after": mean_x, "mean_ndens_after": float(sim.ndens.mean())}
out["speedup_of_the_default"] = out["through_the_host_every_step"]["ms_per_time_step"] / out["class_default"]["ms_per_time_step"]
if out_path:
    with open(out_path, "w") as f:
        f.write(json.dumps(out) + "\n")
print(json.dumps(out))
pc2r.device_close()
