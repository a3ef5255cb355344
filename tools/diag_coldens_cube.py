"""Diagnostic: the whole-cube column densities c2ray_do_all_sources hands back (sub-box kernels) against the oracle's restatement of the
Fortran (bit-identical to the compiled reference on every probe) for ONE source at 256^3: where, and by how much, do they differ?
usage: python tools/diag_coldens_cube.py [R] [workload]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench
import make_fullsize_coldens_golden as MC
from oracle import oracle as O
from pyc2ray_amd.load_extensions import load_c2ray

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
N = 256
thin, thick, dlog = bench.make_tables()
ndens, xh, temp, dr, pos, flux = bench.make_workload(kind, N, 1000)
src = pos[:, -1:]
f = np.asfortranarray
zeros = np.zeros(thin.shape[0])
for opts in ("", "9=1"):
    os.environ["PYC2RAY_AMD_OPTIONS"] = opts
    cd, phi, heat = (np.zeros((N, N, N), order="F") for _ in range(3))
    lib = load_c2ray()
    if opts:
        from pyc2ray_amd.load_extensions import load_asora
        load_asora().set_option(9, 1)
    nbox, loss = lib.raytracing.do_all_sources(flux[-1:], src.astype(np.int32), R, R, cd, bench.SIG, dr, f(ndens), f(xh), phi, heat, 0.0,
                                               thin, thick, zeros, zeros, bench.MINLOGTAU, dlog, float(R))
    ref = O.do_all_sources(flux[-1:], src, R, R, bench.SIG, dr, ndens, xh, 0.0, thin, thick, bench.MINLOGTAU, dlog, float(R),
                           NumTau=thin.shape[0] - 1)["coldens"]
    a, b = MC.cube_of(cd, pos[:, -1], R), MC.cube_of(ref, pos[:, -1], R)
    rel = np.abs(a - b) / np.abs(b)
    rr = np.arange(-R, R + 1)
    dist = np.sqrt(rr[:, None, None] ** 2 + rr[None, :, None] ** 2 + rr[None, None, :] ** 2)
    cheb = np.maximum(np.maximum(np.abs(rr)[:, None, None], np.abs(rr)[None, :, None]), np.abs(rr)[None, None, :])
    print(f"options '{opts}': R={R} {kind}: max rel {rel.max():.3e}, cells > 1e-9: {(rel > 1e-9).sum()} of {rel.size}, > 1e-12: {(rel > 1e-12).sum()}")
    for lo in range(0, R + 1, 8):
        w = (cheb >= lo) & (cheb < lo + 8)
        print(f"   shells {lo:3d}..{lo + 7:3d}: max rel {rel[w].max():.3e}  median {np.median(rel[w]):.3e}")
    worst = np.unravel_index(np.argmax(rel), rel.shape)
    print("   worst cell offset", tuple(int(v) - R for v in worst), "dist %.1f" % dist[worst], "values", a[worst], b[worst])
    big = np.argwhere(rel > 1e-9)
    if len(big):
        off = big - R
        print("   offsets > 1e-9: |di| range", np.abs(off[:, 0]).min(), np.abs(off[:, 0]).max(), "|dj|", np.abs(off[:, 1]).min(), np.abs(off[:, 1]).max(),
              "|dk|", np.abs(off[:, 2]).min(), np.abs(off[:, 2]).max(), "min dist %.1f" % dist[rel > 1e-9].min())
