"""One rank's iteration of the sharded device loop, for a kernel timeline: `python tools/slab_rank_timeline.py [P] [rank] [iterations]`
runs what tools/slab_compute_model.py runs for rank `rank` of `P` (configs[3]) -- meant to be started under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/slab_rank_timeline.py 8 3 8`, then `python tools/timeline.py <csv> 60`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pyc2ray_amd as p
from pyc2ray_amd import _capi
from pyc2ray_amd.dist import SlabPlan, TorchComm
from pyc2ray_amd.load_extensions import load_asora
from pyc2ray_amd.utils.sourceutils import format_sources

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N, R = 256, 32.0
lib = load_asora()
p.device_init(N, 64)
thin, thick, dlog = bench.make_tables()
p.photo_table_to_device(thin, thick)
numtau = thin.shape[0] - 1
ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, 1000)
lib.grid_to_device(_capi.GRID_NDENS, ndens)
lib.grid_to_device(_capi.GRID_TEMP, temp)
lib.grid_to_device(_capi.GRID_XH, xh)
lib.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)
chem = (bench.MYR, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)
spos, sflux, bounds = TorchComm.shard_sources_by_slab(pos, flux, P)
plan = SlabPlan(N, P, R, [spos[0, bounds[q]:bounds[q + 1]] - 1 for q in range(P)])
lo, hi = bounds[r], bounds[r + 1]
p0, f0 = format_sources(spos[:, lo:hi], sflux[lo:hi])
lib.source_data_to_device(p0, f0, hi - lo)
sched, rsched = plan.send_schedule(r, 1), plan.recv_schedule(r, 1)
own_a, own_b = plan.own[r]
halo = [(x, y) for q in range(P) if q != r for x, y in plan.runs[r][q]]
lib.evolve_begin_slab(*chem, R, bench.SIG, dr, bench.MINLOGTAU, dlog, numtau, 0, hi - lo, -1.0, 0.0, own_a, own_b - own_a)
stand_in = lib.device_ptr(_capi.GRID_NDENS)
for it in range(iters + 2):
    lib.evolve_slab_trace(0, hi - lo)
    for _, x, y in sched[0]:
        lib.evolve_slab_fold_out(x, y - x)
    for _, x, y in rsched[0]:
        lib.evolve_slab_add(x, y - x, stand_in + 8 * N * N * x)
    lib.evolve_slab_pass()
    for x, y in halo:
        lib.evolve_slab_nhi(x, y - x)
    lib.evolve_slab_close(None)
    if it == 1:
        lib.evolve_poll(0)
lib.synchronize()
lib.evolve_poll(0)
print("rank", r, "of", P, "sources", hi - lo, "own", plan.own[r], "sends", sched[0], "receives", rsched[0], "halo", halo)
p.device_close()
