"""One raytrace on a FIXED medium -- the quiet field of the headline (xh = 2e-4) and the field with ionisation fronts that
configs[3] x 1e3 fluxes has after one converged time step -- with parts of the kernel switched off through ASORA_ABLATE (read per
call; needs the diagnostic library build/variants/libasora_abl.so in place of the shipped one: results are WRONG while a bit is
set, only the time is of interest).  1 = no rate atomics, 8 = lookups at lane-linear table addresses, 16 = conflict-free log table,
128 = nHI from a 512 KiB window, 64 = rate atomics into a 512 KiB window.
usage (GPU box): python tools/ablate_trace_on_state.py [--ablate 0 1 8 9 16 128]"""
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
ap.add_argument("--ablate", type=int, nargs="+", default=[0, 1, 8, 9, 16, 128])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--flux-scale", type=float, default=1e3)
a = ap.parse_args()
N, ns, R = 256, 1000, 32.0
os.environ["ASORA_ABLATE"] = "0"
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
numtau = thin.shape[0] - 1
ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, ns)
p0, f0 = format_sources(pos, flux * a.flux_scale)
lib.source_data_to_device(p0, f0, ns)
for which, g in ((_capi.GRID_NDENS, ndens), (_capi.GRID_TEMP, temp), (_capi.GRID_XH, xh)):
    lib.grid_to_device(which, g)
chem = (bench.MYR, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)
lib.evolve_begin(*chem, R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, 0, ns, min(int(1e-4 * N ** 3), (ns - 1) / 3), 1e-4)
done = False
while not done:
    lib.evolve_enqueue(4)
    _, done, _ = lib.evolve_poll(8)
x1 = lib.grid_to_host(_capi.GRID_XH_INTERMED, np.empty((N, N, N)))
out = {"ionised_volume_fraction": float((x1 > 0.5).mean()), "rows": []}
for label, field in (("quiet", xh), ("fronts", x1)):
    lib.grid_to_device(_capi.GRID_XH_AV, field)
    for ab in a.ablate:
        os.environ["ASORA_ABLATE"] = str(ab)
        lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, numtau)
        lib.set_option(_capi.OPT_TIMING, 1)
        lib.kernel_time_reset()
        for _ in range(a.reps):
            lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, numtau)
        ms, n = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)
        lib.set_option(_capi.OPT_TIMING, 0)
        out["rows"].append({"medium": label, "ablate": ab, "raytrace_ms": ms / n})
        print(label, "ablate", ab, "raytrace ms", round(ms / n, 4), flush=True)
os.environ["ASORA_ABLATE"] = "0"
p.device_close()
print(json.dumps(out))
