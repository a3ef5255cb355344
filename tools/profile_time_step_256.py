import os, sys, time, collections, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, pyc2ray_amd as p
from pyc2ray_amd.load_extensions import load_asora
N=256
ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, 1000)
ndens=np.asfortranarray(ndens); xh=np.asfortranarray(xh); temp=np.asfortranarray(temp)
lib=load_asora(); p.device_init(N,64)
thin,thick,dlog=bench.make_tables(); p.photo_table_to_device(thin,thick)
spent=collections.defaultdict(float)
def wrap(name, fn):
    def timed(*a, **k):
        t0=time.perf_counter(); r=fn(*a, **k); lib_sync(); spent[name]+=time.perf_counter()-t0; return r
    return timed
lib_sync=lib.synchronize
for name in ("source_data_to_device","grid_to_device","grid_to_host","evolve_begin","evolve_enqueue","evolve_poll"):
    setattr(lib,name,wrap(name,getattr(lib,name)))
def step(x):
    return p.evolve3D(bench.MYR, dr, flux*30, pos, True, 1000, N, 1e-2, temp, ndens, x, thin, thick, bench.MINLOGTAU, dlog, 32.0, 1e-4, bench.SIG, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C, logfile=os.devnull, quiet=True)
x,_=step(xh); spent.clear()
t0=time.perf_counter()
for _ in range(4): x,_=step(x)
tot=time.perf_counter()-t0
print("ms per step", tot/4*1e3, {k: round(v/4*1e3,2) for k,v in spent.items()}, "outside", round((tot-sum(spent.values()))/4*1e3,2))
t0=time.perf_counter(); ndens.mean(); xh.mean(); print("two means ms", (time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); np.empty_like(xh); np.empty((N,N,N)); print("two empties ms", (time.perf_counter()-t0)*1e3)
