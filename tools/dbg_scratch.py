import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
from oracle import oracle as O
import pyc2ray_amd as p
from pyc2ray_amd import _capi as capi
from pyc2ray_amd.load_extensions import load_asora
lib = load_asora()
for N in (104, 168):
    nd, xh, dr = cases.grid(N, "lognormal", 41, 0.02)
    pos, flux = cases.sources(N, 1, 42, flux=5.0)
    thin, thick, dlog = cases.grey_tables()
    if p.cuda_is_init(): p.device_close()
    p.device_init(N, 8); p.photo_table_to_device(thin, thick)
    pos0, fl = cases.flat_sources(pos, flux)
    lib.source_data_to_device(pos0, fl, 1)
    lib.grid_to_device(capi.GRID_NDENS, nd); lib.grid_to_device(capi.GRID_XH_AV, xh)
    ref = O.asora_do_all_sources(1000.0, cases.SIG, dr, nd, xh, pos0, fl, thin, thick, cases.MINLOGTAU, dlog, NumTau=thin.shape[0]-1, flags=O.ASORA_MODE)["phi_ion"]
    for mode in (1, 2):
        for T in (128, 256, 512):
            lib.set_option(capi.OPT_SECTORS, mode); lib.set_option(capi.OPT_BLOCK_THREADS, T)
            lib.raytrace_device(1000.0, cases.SIG, dr, 0, 1, cases.MINLOGTAU, dlog, thin.shape[0]-1)
            phi = lib.grid_to_host(capi.GRID_PHI_ION, np.empty((N,N,N)))
            bad = ~np.isclose(phi, ref, rtol=1e-8, atol=0)
            print(N, "mode", mode, "threads", T, "bad", int(bad.sum()), "of", bad.size, "counts", lib.last_raytrace_counts(), "nz", int((phi!=0).sum()), int((ref!=0).sum()))
