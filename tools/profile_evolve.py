"""Where the wall-clock time of evolve3D goes on a small problem (BASELINE configs[1]: 128^3, one source): every call
into the library is timed on the host (with a device synchronisation behind it, so that the time is attributed to the
call that caused it), over the ten 50 Myr steps of paper test 1.  Prints one JSON line.
usage: python tools/profile_evolve.py [--N 128] [--steps 10]
"""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
import pyc2ray_amd as p
from pyc2ray_amd.load_extensions import load_asora

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--sync", type=int, default=1, help="1: synchronise behind every call (attribution); 0: as in production")
a = ap.parse_args()
N = a.N
myr = 3.15576e13
dr = 5e24 / N
ndens = np.full((N, N, N), 1.87e-7 * (1 + 9.0) ** 3, order="F")
temp = np.full((N, N, N), 1e4, order="F")
src_pos = np.array([[N // 2], [N // 2], [N // 2]])
src_flux = np.array([1e54 / 1e48])
thin, thick, dlog = cases.grey_tables(20000)
colh0, temph0 = 1.3e-8 * 0.83 / 13.598 ** 2, 13.598 / 8.617e-05
chem = (2.59e-13, -0.7, colh0, temph0, 7.1e-7)
R_max_LLS = 15.0 * N / 1.62022035

lib = load_asora()
p.device_init(N, 1)
p.photo_table_to_device(thin, thick)
spent = collections.defaultdict(float)
calls = collections.defaultdict(int)


def wrap(name, fn):
    def timed(*args, **kw):
        t0 = time.perf_counter()
        r = fn(*args, **kw)
        if a.sync and name != "synchronize":
            lib_sync()
        spent[name] += time.perf_counter() - t0
        calls[name] += 1
        return r
    return timed


lib_sync = lib.synchronize
for name in ("source_data_to_device", "grid_to_device", "grid_to_host", "grid_copy", "evolve_begin", "evolve_enqueue",
             "evolve_poll", "synchronize"):
    setattr(lib, name, wrap(name, getattr(lib, name)))


def run(steps):
    xh = np.full((N, N, N), 1.2e-3, order="F")
    iters = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        xh, phi = p.evolve3D(50 * myr, dr, src_flux, src_pos, True, 1000, N, 1e-2, temp, ndens, xh, thin, thick,
                             cases.MINLOGTAU, dlog, R_max_LLS, 1e-4, cases.SIG, *chem, logfile=os.devnull, quiet=True)
        iters += p.evolve._evolve.last_niter
    return time.perf_counter() - t0, iters


run(1)
spent.clear(); calls.clear()
total, iters = run(a.steps)
inside = sum(spent.values())
print(json.dumps({"case": f"paper test 1, {N}^3, one source, {a.steps} steps of 50 Myr", "seconds": total, "outer_iterations": iters,
                  "ms_per_iteration": total / iters * 1e3, "synchronised_after_every_call": bool(a.sync),
                  "ms_per_time_step_by_library_call": {k: round(v / a.steps * 1e3, 3) for k, v in sorted(spent.items(), key=lambda kv: -kv[1])},
                  "calls_per_time_step": {k: calls[k] / a.steps for k in spent},
                  "ms_per_time_step_outside_the_library": round((total - inside) / a.steps * 1e3, 3)}))
p.device_close()
