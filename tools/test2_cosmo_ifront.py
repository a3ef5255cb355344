"""The reference's paper test 2 (test/paper_tests/test2_Ifront_cosmo): ionisation front of one source of 1e54 photons/s
in an expanding uniform medium (n_H = 1.87e-7 cm^-3 comoving, z = 9 onwards, box 22.685 comoving Mpc, 256^3, grey
opacity), ten steps of 50 Myr ("coarse" mode of run_test.py), through C2Ray_Test with cosmology switched on.  The
front radius (x = 0.5 along +i from the source, in comoving kpc / 10 as make_plot.ipynb cells 1,4,7 take it) is compared
with the analytic solution of make_plot.ipynb cell 5 (Shapiro & Giroux form with E_2); the reference's figure shows
r_N / r_A within [0.985, 1.005].  Prints one JSON line.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
from scipy.special import expn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyc2ray_amd as pc2r
from pyc2ray_amd.c2ray_base import FlatLambdaCDMLite

N = 256
BASE = open(os.path.join(ROOT, "tests", "data", "parameters_single_black_body.yml")).read()
txt = (BASE.replace("boxsize: 0.014", "boxsize: 22.685455026110553").replace("avg_dens: 1.0e-6", "avg_dens: 1.87e-7")
           .replace("NumTau: 10000", "NumTau: 20000").replace("grey: 0", "grey: 1")
           .replace("R_max_cMpc: 0.01640625", "R_max_cMpc: 15.0").replace("cosmological: 0", "cosmological: 1")
           .replace("h: 1.0", "h: 0.7").replace("Omega_B: 0.044", "Omega_B: 0.043").replace("subboxsize: 150", "subboxsize: 128"))
work = tempfile.mkdtemp()
os.chdir(work)
open("parameters.yml", "w").write(txt)
open("source.txt", "w").write("1\n128 128 128 1e54 0.0\n")

real_stdout = os.dup(1)
os.dup2(2, 1)
sim = pc2r.C2Ray_Test("parameters.yml", N, True)
numzred, t_evol = 10, 5e8
zred_array = sim.generate_redshift_array(numzred + 1, t_evol / numzred)
srcpos, srcflux = sim.read_sources("source.txt", 1)
profiles = [np.array(sim.xh[127:, 127, 127])]
t0 = time.perf_counter()
iters = 0
for k in range(numzred):
    zi, zf = zred_array[k], zred_array[k + 1]
    dt = sim.set_timestep(zi, zf, 1)
    sim.zred = zi
    sim.set_constant_average_density(1.87e-7, zi)
    sim.cosmo_evolve(dt)
    sim.evolve3D(dt, srcflux, srcpos)
    iters += pc2r.evolve._evolve.last_niter
    profiles.append(np.array(sim.xh[127:, 127, 127]))
secs = time.perf_counter() - t0
pc2r.device_close()

# analytic solution, make_plot.ipynb cell 5
kpc, year = 3.086e21, 3.15576e7
cosmo = FlatLambdaCDMLite(70, 0.27, 2.726, Ob0=0.043)
ti = cosmo.age(9) / (1e6 * year)
r_S = ((3 * 1e54) / (4 * np.pi * 2.59e-13 * 1.87e-4 ** 2)) ** (1. / 3) / kpc
t_rec = 1.0 / (2.59e-13 * 1.87e-4 * year * 1e6)
lam = ti / t_rec
y = lambda t: lam * np.exp(lam * ti / t) * (t / ti * expn(2, lam * ti / t) - expn(2, lam))
r_I = lambda t: r_S * y(ti + t) ** (1. / 3)
x = np.linspace(0, 22685 / 10 / 2, N // 2 + 1)
front = [float(np.interp(0.5, np.flip(p), np.flip(x))) for p in profiles[1:]]
tt = np.linspace(0, 500, 11)[1:]
ratios = [f / float(r_I(t)) for f, t in zip(front, tt)]
sys.stdout.flush()
os.dup2(real_stdout, 1)
print(json.dumps({"case": "paper test 2 (cosmological I-front), 256^3, ten 50 Myr steps", "seconds": secs,
                  "outer_iterations": iters, "t_i_Myr": ti, "lambda": lam, "front_kpc": front,
                  "analytic_kpc": [float(r_I(t)) for t in tt], "r_N_over_r_A": [round(r, 4) for r in ratios],
                  "final_redshift": float(sim.zred), "reference_figure_band": [0.985, 1.005]}))
