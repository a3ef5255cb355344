"""The reference's only self-checking regression (test/unit_tests_hackathon/1_single_black_body/run_test.py): 128^3,
one 5e4 K black-body source of 1e49 photons/s at (96,96,64), uniform n = 1e-3 cm^-3, ten steps of 1 Myr, driven through
the C2Ray_Test class exactly as that script drives it; the final ionised fraction is compared per cell with a golden
field under the script's eight thresholds (run_test.py:91-115).

The script's golden file (original C2-Ray output) is not part of the reference checkout.  The golden field here is
produced by the reference's own Fortran (oracle/_ref, CPU, one core) driven by the same loop -- the reference's
use_gpu=False path; the candidate is this build's ASORA path on the GPU (what `run_test.py --gpu` runs).
Prints one JSON line.  usage: python tools/hackathon_test1.py [--steps 10]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyc2ray_amd as pc2r

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--shadow", action="store_true",
                help="paper test 4 instead (test/paper_tests/test4_shadow/shadow.py): source at the centre, a clump of 6x "
                     "the density and radius 8 cells at (76,76,63) casting a shadow, C-ordered density, subboxsize 64")
ap.add_argument("--N", type=int, default=128)
a = ap.parse_args()
PARAMS = os.path.join(ROOT, "tests", "data", "parameters_single_black_body.yml")
N = a.N
work = tempfile.mkdtemp()
os.chdir(work)
with open("src.txt", "w") as f:
    f.write(f"1\n{N // 2} {N // 2} {N // 2} 10e48 1.0\n" if a.shadow else f"1\n{3 * N // 4} {3 * N // 4} {N // 2} 10e48 0.0\n")
if a.shadow:
    txt = open(PARAMS).read().replace("R_max_cMpc: 0.01640625", "R_max_cMpc: 15.0").replace("subboxsize: 150", "subboxsize: 64")
    PARAMS = os.path.join(work, "parameters.yml")
    open(PARAMS, "w").write(txt)


def density():
    nd = 1e-3 * np.ones((N, N, N))                      # C-ordered, as the reference's scripts build it
    if a.shadow:
        i, j, k = np.ogrid[0:N, 0:N, 0:N]
        c, r = (np.array([76, 76, 63]) * N) // 128, 8 * N // 128
        nd[(i - c[0]) ** 2 + (j - c[1]) ** 2 + (k - c[2]) ** 2 < r * r] = 6e-3
    return nd


def drive(use_gpu):
    sim = pc2r.C2Ray_Test(PARAMS, N, use_gpu)
    zs = sim.generate_redshift_array(2, 1e7)
    srcpos, srcflux = sim.read_sources("src.txt", 1)
    sim.ndens = density()
    dt = sim.set_timestep(zs[0], zs[1], 10)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sim.cosmo_evolve(dt)
        sim.evolve3D(dt, srcflux, srcpos)
    return sim, np.array(sim.xh), time.perf_counter() - t0, dt, srcpos, srcflux


real_stdout = os.dup(1)
os.dup2(2, 1)                                               # the class logs to stdout: keep it for the JSON line
sim, x_gpu, t_gpu, dt, srcpos, srcflux = drive(True)
pc2r.device_close()
_, x_sub, t_sub, *_ = drive(False)

out = {"case": (f"paper test 4 (shadow behind a dense clump) at {N}^3, {a.steps} steps of 1 Myr" if a.shadow else
                f"unit_tests_hackathon/1_single_black_body at {N}^3, {a.steps} steps of 1 Myr"),
       "mean_x_asora_path": float(x_gpu.mean()), "seconds_asora_path": t_gpu,
       "mean_x_subbox_semantics_path": float(x_sub.mean()), "seconds_subbox_semantics_path": t_sub}

from oracle import ref_fortran as F
if F.available():
    chem = (sim.bh00, sim.albpow, sim.colh0, sim.temph0, sim.abu_c)
    ndens = np.asfortranarray(sim.ndens)
    temp = np.asfortranarray(sim.temp)
    xh = np.full((N, N, N), 1.2e-3, order="F")
    t0 = time.perf_counter()
    iters = 0
    for _ in range(a.steps):                                # pyc2ray/evolve.py:116-245, use_gpu=False
        xh_av, xh_int = xh.copy(order="F"), xh.copy(order="F")
        prev1 = prev0 = 2 * N ** 3
        while True:
            iters += 1
            r = F.do_all_sources(srcflux, srcpos, max_subbox=sim.max_subbox, subboxsize=sim.subboxsize, sig=sim.sig,
                                 dr=sim.dr, ndens=ndens, xh_av=xh_av, loss_fraction=sim.loss_fraction,
                                 thin=sim.photo_thin_table, thick=sim.photo_thick_table, minlogtau=sim.minlogtau,
                                 dlogtau=sim.dlogtau, R_max_LLS=sim.R_max_LLS)
            xh_av, xh_int, conv = F.global_pass(dt, ndens, temp, xh, xh_av, xh_int, r["phi_ion"], *chem)
            s1, s0 = np.sum(xh_int), np.sum(1.0 - xh_int)
            rel1, rel0 = abs((s1 - prev1) / s1), abs((s0 - prev0) / s0)
            prev1, prev0 = s1, s0
            if conv < min(int(sim.convergence_fraction * N ** 3), 0) or (rel1 < sim.convergence_fraction
                                                                           and rel0 < sim.convergence_fraction):
                break
        xh = xh_int
    t_cpu = time.perf_counter() - t0
    out.update({"mean_x_reference_fortran": float(xh.mean()), "seconds_reference_fortran_1_core": t_cpu,
                "outer_iterations_reference": iters})
    for tag, cand in (("asora_path_vs_reference", x_gpu), ("subbox_semantics_path_vs_reference", x_sub)):
        abserr = cand - xh
        relerr = abserr / xh
        stats = {"abs_mean": abserr.mean(), "abs_std": abserr.std(), "abs_max": abserr.max(), "abs_min": abserr.min(),
                 "rel_mean": relerr.mean(), "rel_std": relerr.std(), "rel_max": relerr.max(), "rel_min": relerr.min()}
        limits = {"abs_mean": 1e-8, "abs_std": 3e-7, "abs_max": 5e-6, "abs_min": 5e-6,        # run_test.py:91-101
                  "rel_mean": 1e-7, "rel_std": 3e-6, "rel_max": 2e-5, "rel_min": 2e-5}        # run_test.py:105-115
        out[tag] = {k: float(v) for k, v in stats.items()}
        out[tag]["failed_thresholds"] = [k for k in stats if abs(stats[k]) > limits[k]]
sys.stdout.flush()
os.dup2(real_stdout, 1)
print(json.dumps(out))
