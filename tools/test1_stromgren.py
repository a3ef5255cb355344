"""BASELINE configs[1]: the reference's paper test 1 (test/paper_tests/test1_Ifront) -- one source of 1e54 photons/s,
uniform n_H = 1.87e-4 cm^-3, T = 1e4 K, grey opacity, box 5e24 cm, 128^3 cells, ten steps of 50 Myr -- on the MI355X
(evolve3D with use_gpu=True: ASORA path; use_gpu=False: sub-box semantics), timed, with the ionisation-front radius
against the analytic solution, and the FIRST time step repeated with the reference Fortran (oracle/_ref, one core)
driven by the same loop, for time and for the difference of the resulting ionised fractions.
Prints one JSON line.  usage: python tools/test1_stromgren.py [--cpu-steps 1]
"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--cpu-steps", type=int, default=1)
ap.add_argument("--N", type=int, default=128)
a = ap.parse_args()

N = a.N
kpc, myr = 3.086e21, 3.15576e13
dr = 5e24 / N
ndens = np.full((N, N, N), 1.87e-7 * (1 + 9.0) ** 3, order="F")
temp = np.full((N, N, N), 1e4, order="F")
src_pos = np.array([[N // 2], [N // 2], [N // 2]])
src_flux = np.array([1e54 / 1e48])
thin, thick, dlog = cases.grey_tables(20000)
colh0, temph0 = 1.3e-8 * 0.83 / 13.598 ** 2, 13.598 / 8.617e-05
R_max_LLS = 15.0 * N / 1.62022035
r_S = ((3 * 1e54) / (4 * np.pi * 2.59e-13 * 1.87e-4 ** 2)) ** (1. / 3) / kpc
t_rec = 1.0 / (2.59e-13 * 1.87e-4 * myr)
x_axis = (np.arange(N - (N // 2 - 1)) * dr) / kpc
chem = (2.59e-13, -0.7, colh0, temph0, 7.1e-7)


def run(use_gpu, steps):
    xh = np.full((N, N, N), 1.2e-3, order="F")
    ratios, iters, t0 = [], 0, time.perf_counter()
    for step in range(1, steps + 1):
        xh, phi = p.evolve3D(50 * myr, dr, src_flux, src_pos, use_gpu, 1000, N, 1e-2, temp, ndens, xh, thin, thick,
                             cases.MINLOGTAU, dlog, R_max_LLS, 1e-4, cases.SIG, *chem, logfile=os.devnull, quiet=True)
        iters += p.evolve._evolve.last_niter
        prof = xh[N // 2 - 1:, N // 2 - 1, N // 2 - 1]
        front = np.interp(0.5, np.flip(prof), np.flip(x_axis))
        ratios.append(front / (r_S * (1.0 - np.exp(-50.0 * step / t_rec)) ** (1. / 3)))
    return xh, time.perf_counter() - t0, iters, ratios


p.device_init(N, 1)
p.photo_table_to_device(thin, thick)
run(True, 1)                                            # warm-up (geometry tables, first-touch)
xh_gpu, t_gpu, it_gpu, ratios = run(True, 10)
xh_sub, t_sub, it_sub, ratios_sub = run(False, 10)
out = {"case": f"BASELINE configs[1]: test 1 (Stroemgren sphere), {N}^3, one source, ten 50 Myr steps",
       "asora_path": {"seconds": t_gpu, "outer_iterations": it_gpu, "ms_per_iteration": 1e3 * t_gpu / it_gpu,
                      "front_radius_over_analytic": [round(float(r), 4) for r in ratios]},
       "subbox_semantics_path": {"seconds": t_sub, "outer_iterations": it_sub, "ms_per_iteration": 1e3 * t_sub / it_sub,
                                 "front_radius_over_analytic": [round(float(r), 4) for r in ratios_sub],
                                 "max_abs_diff_of_x_vs_asora_path": float(np.abs(xh_sub - xh_gpu).max())}}

if a.cpu_steps > 0:
    from oracle import ref_fortran as F
    if F.available():
        xh1_gpu, _, _, _ = run(False, a.cpu_steps)
        xh = np.full((N, N, N), 1.2e-3, order="F")
        t0 = time.perf_counter()
        iters = 0
        for step in range(a.cpu_steps):                 # the use_gpu=False loop of pyc2ray/evolve.py:168-245
            xh_av, xh_int = xh.copy(order="F"), xh.copy(order="F")
            prev1 = prev0 = 2 * N ** 3
            while True:
                iters += 1
                r = F.do_all_sources(src_flux, src_pos, max_subbox=1000, subboxsize=N, sig=cases.SIG, dr=dr, ndens=ndens,
                                     xh_av=xh_av, loss_fraction=1e-2, thin=thin, thick=thick, minlogtau=cases.MINLOGTAU,
                                     dlogtau=dlog, R_max_LLS=R_max_LLS)
                xh_av, xh_int, conv = F.global_pass(50 * myr, ndens, temp, xh, xh_av, xh_int, r["phi_ion"], *chem)
                s1, s0 = np.sum(xh_int), np.sum(1.0 - xh_int)
                rel1, rel0 = abs((s1 - prev1) / s1), abs((s0 - prev0) / s0)
                prev1, prev0 = s1, s0
                if conv < min(int(1e-4 * N ** 3), 0) or (rel1 < 1e-4 and rel0 < 1e-4):
                    break
            xh = xh_int
        t_cpu = time.perf_counter() - t0
        w = xh > 1e-3
        out["reference_fortran_cpu"] = {"steps": a.cpu_steps, "cores": 1, "seconds": t_cpu, "outer_iterations": iters,
                                        "seconds_per_iteration": t_cpu / iters,
                                        "max_rel_diff_of_x_vs_subbox_semantics_path_on_the_gpu":
                                            float(np.max(np.abs(xh1_gpu[w] - xh[w]) / xh[w]))}
print(json.dumps(out))
p.device_close()
