"""PCIe-inclusive rate of the drop-in call libasora.do_all_sources (host xh_av in, host phi_ion out),
the way the reference's raytracing benchmark times it (raytracing_benchmark/run_test.py:82-91).
Never used as the bench `value`; reported in DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pyc2ray_amd as p
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

N, ns, R = 256, 1000, 32.0
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
p0, f0 = format_sources(pos, flux)
lib.source_data_to_device(p0, f0, ns)
nd_flat = np.ravel(ndens).astype("float64", copy=True)
lib.density_to_device(nd_flat, N)
xh_flat = np.ravel(xh).astype("float64", copy=True)
phi_flat = np.zeros(N ** 3)
cd_flat = np.zeros(N ** 3)
out = {"call": "libasora.do_all_sources (H2D xh_av + raytrace + D2H phi_ion)", "N": N, "sources": ns, "R": R}
for label, opt in (("pipelined_copies", 1), ("copies_in_turn", 0)):
    lib.set_option(10, opt)               # ASORA_OPT_PIPELINED_COPIES
    for _ in range(2):
        lib.do_all_sources(R, cd_flat, bench.SIG, dr, nd_flat, xh_flat, phi_flat, ns, N, bench.MINLOGTAU, dlog, thin.shape[0] - 1)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.do_all_sources(R, cd_flat, bench.SIG, dr, nd_flat, xh_flat, phi_flat, ns, N, bench.MINLOGTAU, dlog, thin.shape[0] - 1)
    t = (time.perf_counter() - t0) / reps
    gam, ev = lib.last_raytrace_counts()
    out[label] = {"s_per_call": t, "raytrace_cell_updates_per_s": gam / t,
                  "ns_per_source_per_insphere_cell": t * 1e9 / (ns * 4 * np.pi * R ** 3 / 3), "checksum": float(phi_flat.sum())}
lib.set_option(10, 1)
print(json.dumps(out))
p.device_close()
