"""Deterministic synthetic inputs shared by the golden generator, the CPU tests and the GPU
parity tests.  Everything is seeded; nothing here reads /root/reference."""
import numpy as np

SIG = 6.30e-18          # sigma_HI_at_ion_freq (test/*/parameters.yml)
MINLOGTAU, MAXLOGTAU = -20.0, 4.0

# chemistry constants as derived in pyc2ray/c2ray_base.py:335-347 from parameters.yml
BH00 = 2.59e-13
ALBPOW = -0.7
COLH0 = 1.3e-8 * 0.83 * 1.0 / 13.598 ** 2
TEMPH0 = 13.598 * 11604.5250061598
ABU_C = 7.1e-7


def tau_table(num_tau):
    """make_tau_table convention (pyc2ray/radiation/common.py:13-37): tau[0]=0 then log-spaced."""
    dlog = (MAXLOGTAU - MINLOGTAU) / num_tau
    tau = np.empty(num_tau + 1)
    tau[0] = 0.0
    tau[1:] = 10 ** (MINLOGTAU + np.arange(num_tau) * dlog)
    return tau, dlog


def grey_tables(num_tau=2000):
    """Closed-form grey-opacity tables: thick = thin = 1e48*exp(-tau).  Any monotone table pins
    the kernels, since the checker and the kernel receive the same table."""
    tau, dlog = tau_table(num_tau)
    t = 1e48 * np.exp(-tau)
    return t.copy(), t.copy(), dlog


def soft_tables(num_tau=2000):
    """A non-grey-like pair (power-law tail), to make thin != thick."""
    tau, dlog = tau_table(num_tau)
    thick = 1e48 * (0.6 * np.exp(-tau) + 0.4 / (1.0 + tau) ** 3)
    thin = 1e48 * (0.6 * np.exp(-tau) + 1.2 / (1.0 + tau) ** 4)
    return thin, thick, dlog


def grid(N, kind, seed, tau_cell, xlo=1e-4, xhi=0.5):
    """Return (ndens, xh, dr).  tau_cell = sig * 1e-3 * dr, the optical depth of a mean cell."""
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        nd = np.full((N, N, N), 1e-3)
        xh = np.full((N, N, N), 2e-4)
    elif kind == "lognormal":
        nd = 1e-3 * np.exp(1.2 * rng.normal(size=(N, N, N)) - 0.72)
        xh = rng.uniform(xlo, xhi, size=(N, N, N))
    else:
        raise ValueError(kind)
    dr = tau_cell / (SIG * 1e-3)
    return nd, xh, dr


def sources(N, ns, seed, flux=1.0):
    """(3,ns) 1-based positions as in generate_test_sourcefile (sourceutils.py:56-58), equal flux."""
    rng = np.random.RandomState(seed)
    pos = 1 + rng.randint(0, N, size=3 * ns)
    pos = pos.reshape((ns, 3), order="C").T.copy()
    return pos, np.full(ns, float(flux))


def flat_sources(pos1, flux):
    """format_sources layout (sourceutils.py:30-31): 0-based, xyz-interleaved int32."""
    return (np.ravel((pos1 - 1).astype("int32"), order="F"), flux.astype("float64"))


#: raytracing cases: name -> (N, kind, grid_seed, tau_cell, ns, src_seed, R)
RT_CASES = {
    "u16_1src_R4":   (16, "uniform",   0, 0.30, 1, 11, 4.0),
    "u16_1src_R8":   (16, "uniform",   0, 0.30, 1, 11, 8.0),
    "u16_7src_Rbox": (16, "uniform",   0, 0.05, 7, 12, 1000.0),
    "l16_7src_R5.5": (16, "lognormal", 3, 0.08, 7, 13, 5.5),
    "l17_3src_Rbox": (17, "lognormal", 4, 0.05, 3, 14, 1000.0),
    "l32_5src_R10":  (32, "lognormal", 5, 0.10, 5, 15, 10.0),
    "l16_thin":      (16, "lognormal", 6, 1e-9, 3, 16, 1000.0),
    "l16_thick":     (16, "lognormal", 7, 30.0, 2, 17, 1000.0),
}


def rt_case(name, tables="grey"):
    N, kind, gseed, tau_cell, ns, sseed, R = RT_CASES[name]
    nd, xh, dr = grid(N, kind, gseed, tau_cell)
    pos, flux = sources(N, ns, sseed, flux=3.0)
    thin, thick, dlog = grey_tables() if tables == "grey" else soft_tables()
    return dict(N=N, ndens=nd, xh=xh, dr=dr, pos=pos, flux=flux, R=R,
                thin=thin, thick=thick, dlogtau=dlog, minlogtau=MINLOGTAU, sig=SIG)


def chem_case(N, seed, dt=3.15576e13):
    """Inputs of one global_pass: lognormal density, mixed ionisation, Gamma spanning 0..huge."""
    rng = np.random.default_rng(seed)
    nd = 1e-3 * np.exp(1.0 * rng.normal(size=(N, N, N)))
    temp = 10 ** rng.uniform(3.0, 4.7, size=(N, N, N))
    xh = 10 ** rng.uniform(-4, -0.01, size=(N, N, N))
    xh_av = np.clip(xh * rng.uniform(0.5, 1.5, size=(N, N, N)), 1e-10, 1 - 1e-10)
    phi = 10 ** rng.uniform(-20, -8, size=(N, N, N))
    phi[rng.uniform(size=(N, N, N)) < 0.2] = 0.0
    phi[rng.uniform(size=(N, N, N)) < 0.02] = 1e-2     # -> fully ionised cells
    return dict(dt=dt, ndens=nd, temp=temp, xh=xh, xh_av=xh_av, xh_intermed=xh.copy(), phi_ion=phi,
                bh00=BH00, albpow=ALBPOW, colh0=COLH0, temph0=TEMPH0, abu_c=ABU_C)


#: sub-box cases of the reference's CPU raytracer (libc2ray.raytracing.do_all_sources): name ->
#: (raytracing case, tables, max_subbox, subboxsize, loss_fraction, unequal fluxes?[, tau of a mean cell])
#: All with R_max_LLS = 1000 cells: with a smaller radius, cells on the box faces deposit nothing and the
#: reference adds an UNDEFINED phi_out to the loss there (raytracing.f90:408,519,543).
SUBBOX_CASES = {
    "sb32_b5":   ("l32_5src_R10", "soft", 1000, 5, 0.05, True),
    "sb32_b3":   ("l32_5src_R10", "grey", 12, 3, 1e-2, False),
    "sb17_b2":   ("l17_3src_Rbox", "soft", 6, 2, 0.3, True),
    "sb16_thick": ("l16_thick", "soft", 1000, 2, 1e-2, False),
    "sb16_b7":   ("l16_7src_R5.5", "soft", 1000, 7, 1e-3, True),
    "sb16_b8":   ("u16_7src_Rbox", "grey", 1000, 8, 1e-3, False),
    "sb32_mid":  ("l32_5src_R10", "soft", 1000, 3, 1e-2, True, 1.0),     # sources stop after different box counts
    "sb32_mid2": ("l32_5src_R10", "grey", 1000, 4, 0.1, True, 0.5),
    "sb17_mid":  ("l17_3src_Rbox", "soft", 1000, 1, 0.05, False, 1.5),
}


def subbox_case(name):
    """Inputs of a sub-box case: the raytracing case + heating tables + (optionally) unequal fluxes."""
    base, tables, max_subbox, subboxsize, loss_fraction, unequal = SUBBOX_CASES[name][:6]
    c = rt_case(base, tables)
    c["R"] = 1000.0
    if len(SUBBOX_CASES[name]) > 6:
        c["dr"] = SUBBOX_CASES[name][6] / (SIG * 1e-3)
    n = c["thin"].shape[0]
    c["heat_thin"] = 1e-11 * c["thin"] * np.linspace(1.0, 2.0, n)
    c["heat_thick"] = 0.7e-11 * c["thick"] * np.linspace(2.0, 1.0, n)
    if unequal:
        c["flux"] = c["flux"] * (0.5 + np.arange(c["flux"].shape[0]))
    c.update(max_subbox=max_subbox, subboxsize=subboxsize, loss_fraction=loss_fraction)
    return c


# ---- evolve3D / do_raytracing cases pinned by the reference's own Python (tests/golden/make_evolve_golden.py) ------
def blackbody_tables(teff=5e4, num_tau=2000):
    """The black-body tables of the reference's tests (parameters.yml: Teff, cross_section_pl_index 2.8,
    integration from the HI threshold to 10 x the HeII threshold), from this package's table builder.  Both sides of
    every comparison receive the same arrays."""
    from pyc2ray_amd.radiation import BlackBodySource, make_tau_table
    ev2fr = 0.241838e15
    tau, dlog = make_tau_table(MINLOGTAU, MAXLOGTAU, num_tau)
    src = BlackBodySource(teff, False, ev2fr * 13.598, 2.8)
    thin, thick = src.make_photo_table(tau, ev2fr * 13.598, 10 * ev2fr * 54.416, 1e48)
    return thin, thick, dlog


MYR = 3.15576e13
#: name -> dict(N, grid kind/seed/tau_cell or "cfg0", sources, flux, R, use_gpu, order, dt, steps, ...)
EVOLVE_CASES = {
    # BASELINE.json configs[0], verbatim: one point source, uniform 64^3 density, r_RT = 32, the Fortran CPU path
    # (ndens 1e-3, xh 1.2e-3, T 1e4, source (32,32,32), normflux 1e6; box, time step and sub-box
    # parameters of unit_tests_hackathon/1_single_black_body/parameters.yml)
    "cfg0_64":        dict(kind="cfg0", N=64, use_gpu=False, order="F", steps=2),
    "l16_cpu_F":      dict(kind="lognormal", N=16, seed=31, tau_cell=0.2, ns=3, sseed=41, flux=2e-4, R=1000.0, use_gpu=False,
                           order="F", steps=2, dt=2 * MYR, subboxsize=3, max_subbox=1000, loss_fraction=1e-2),
    "l24_cpu_F_7src": dict(kind="lognormal", N=24, seed=32, tau_cell=0.1, ns=7, sseed=42, flux=4e-5, R=1000.0, use_gpu=False,
                           order="F", steps=1, dt=3 * MYR, subboxsize=5, max_subbox=1000, loss_fraction=0.05),
    # (the reference's loop only accepts Fortran-ordered xh: its np.copy(xh) goes to an f2py intent(inout) argument)
    "l16_gpu_F":      dict(kind="lognormal", N=16, seed=33, tau_cell=0.2, ns=4, sseed=43, flux=1e-4, R=6.5, use_gpu=True,
                           order="F", steps=2, dt=2 * MYR),
    "l24_gpu_F_37src": dict(kind="lognormal", N=24, seed=34, tau_cell=0.15, ns=37, sseed=44, flux=1.6e-5, R=9.0, use_gpu=True,
                            order="F", steps=1, dt=3 * MYR),
    "l32_gpu_F_5src": dict(kind="lognormal", N=32, seed=35, tau_cell=0.15, ns=5, sseed=45, flux=2.8e-4, R=11.0, use_gpu=True,
                           order="F", steps=1, dt=3 * MYR),
}
RAYTRACING_CASES = ("l16_cpu_F", "l24_cpu_F_7src")


def evolve_case(name):
    s = EVOLVE_CASES[name]
    N = s["N"]
    if s["kind"] == "cfg0":
        nd = np.full((N, N, N), 1e-3)
        xh = np.full((N, N, N), 1.2e-3)
        dr = 0.014 * 3.086e24 / N
        pos = np.array([[32], [32], [32]])
        flux = np.array([1e6])
        thin, thick, dlog = blackbody_tables()
        c = dict(R=32.0, dt=MYR, subboxsize=150, max_subbox=1000, loss_fraction=1e-2)
    else:
        nd, xh, dr = grid(N, s["kind"], s["seed"], s["tau_cell"], xlo=1e-4, xhi=2e-3)
        pos, flux = sources(N, s["ns"], s["sseed"], flux=s["flux"])
        thin, thick, dlog = soft_tables()
        c = dict(R=s["R"], dt=s["dt"], subboxsize=s.get("subboxsize", N), max_subbox=s.get("max_subbox", 1000),
                 loss_fraction=s.get("loss_fraction", 1e-2))
    order = s["order"]
    n = thin.shape[0]
    c.update(N=N, ndens=np.asarray(nd, order=order), xh=np.asarray(xh, order=order),
             temp=np.asarray(np.full((N, N, N), 1e4), order=order), dr=dr, pos=pos, flux=flux, thin=thin, thick=thick,
             dlogtau=dlog, use_gpu=s["use_gpu"], order=order, steps=s["steps"], convergence_fraction=1e-4,
             heat_thin=1e-11 * thin * np.linspace(1.0, 2.0, n), heat_thick=0.7e-11 * thick * np.linspace(2.0, 1.0, n))
    return c
