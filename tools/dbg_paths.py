import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources
N, ns, R = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
lib = load_asora()
thin, thick, dlog = bench.make_tables()
ndens, xh, temp, dr, pos, flux = bench.make_workload("uniform", N, ns)
lib.device_init_auto(N)
lib.photo_table_to_device(thin, thick, thin.shape[0])
p0, f0 = format_sources(pos, flux)
lib.source_data_to_device(p0, f0, ns)
lib.grid_to_device(0, ndens); lib.grid_to_device(1, xh)
lib.set_option(0, 1)
res = {}
for mode, thr in ((0, 0), (1, 256), (2, 512), (2, 1024), (3, 512)):
    lib.set_option(5, mode); lib.set_option(4, thr)
    lib.raytrace_device(R, bench.SIG, dr, 0, ns, bench.MINLOGTAU, dlog, thin.shape[0])
    res[(mode, thr)] = lib.grid_to_host(2, np.empty((N, N, N)))
lib.set_option(5, 0); lib.set_option(4, 0); lib.set_option(0, 0)
nbox, loss = lib.subbox_raytrace_device(int(R), int(R), 0.0, R, bench.SIG, dr, bench.MINLOGTAU, dlog, thin.shape[0], 0, ns)
phi_s = lib.grid_to_host(2, np.empty((N, N, N)))
ref = O.do_all_sources(flux, pos, 1000, N, bench.SIG, dr, ndens, xh, 0.0, thin, thick, bench.MINLOGTAU, dlog, R)["phi_ion"]
ref = np.ascontiguousarray(ref)
w = ref != 0
def md(a): 
    d = np.abs(a[w] - ref[w]) / ref[w]; i = np.argmax(d); return float(d.max()), float(ref[w][i]), float(a[w][i])
print("subbox vs oracle", md(phi_s))
for k, v in res.items(): print("asora", k, "vs oracle", md(v), "min phi", float(ref[w].min()), "max", float(ref.max()))
a = res[(0, 0)]
print("support equal: asora/oracle", bool(np.array_equal(a != 0, ref != 0)), "subbox/oracle", bool(np.array_equal(phi_s != 0, ref != 0)))
print("negatives: oracle", int((ref < 0).sum()), "asora", int((a < 0).sum()), "subbox", int((phi_s < 0).sum()))
print("oracle min positive", float(ref[ref > 0].min()), "asora min positive", float(a[a > 0].min()), "subbox min abs", float(np.abs(phi_s[phi_s != 0]).min()))
for nm, g in (("asora", a), ("subbox", phi_s)):
    rel = np.abs(g[w] - ref[w]) / np.abs(ref[w])
    print(nm, "rel-diff quantiles vs oracle", [float(q) for q in np.quantile(rel, [0.5, 0.99, 0.9999, 1.0])])
idx = np.unravel_index(np.argmin(np.where(a > 0, a, np.inf)), a.shape)
print("cell of smallest asora value", idx, "oracle", float(ref[idx]), "asora", float(a[idx]), "subbox", float(phi_s[idx]), "sources", pos.T.tolist())
