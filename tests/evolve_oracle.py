"""TEST INFRASTRUCTURE: the reference's evolve3D loop (pyc2ray/evolve.py:116-245, GPU branch)
restated on top of the CPU oracle, to check pyc2ray_amd.evolve3D end to end."""
import numpy as np

from oracle import oracle as O


def evolve3D_oracle(dt, dr, src_flux, src_pos, temp, ndens, xh, thin, thick, minlogtau, dlogtau, R_max_LLS,
                    convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, flags=O.ASORA_MODE,
                    max_iter=100):
    NumSrc = src_flux.shape[0]
    N = temp.shape[0]
    NumCells = N ** 3
    NumTau = thin.shape[0]                                     # evolve.py:124
    conv_criterion = min(int(convergence_fraction * NumCells), (NumSrc - 1) / 3)
    prev1 = prev0 = 2 * NumCells
    xh_av = np.array(xh, dtype=np.float64, order="C", copy=True)
    xh_intermed = xh_av.copy()
    pos0 = np.ravel((np.asarray(src_pos) - 1).astype("int32"), order="F")
    history = []
    converged = False
    niter = 0
    phi = None
    while not converged and niter < max_iter:
        niter += 1
        phi = O.asora_do_all_sources(R_max_LLS, sig, dr, ndens, xh_av, pos0, src_flux, thin, thick,
                                     minlogtau, dlogtau, NumTau=NumTau, flags=flags)["phi_ion"]
        xh_av, xh_intermed, conv_flag, _ = O.global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi,
                                                         bh00, albpow, colh0, temph0, abu_c)
        s1 = np.sum(xh_intermed)
        s0 = np.sum(1.0 - xh_intermed)
        rel1 = abs((s1 - prev1) / s1) if s1 > 0 else 1.0
        rel0 = abs((s0 - prev0) / s0) if s0 > 0 else 1.0
        history.append((conv_flag, rel1, rel0))
        converged = (conv_flag < conv_criterion) or (rel1 < convergence_fraction and rel0 < convergence_fraction)
        prev1, prev0 = s1, s0
    return xh_intermed, phi, niter, history


def evolve3d_cpu_path(dt, dr, src_flux, src_pos, max_subbox, subboxsize, loss_fraction, temp, ndens, xh, thin, thick,
                      minlogtau, dlogtau, R, conv, sig, max_iter=100):
    """The use_gpu=False branch (pyc2ray/evolve.py:168-245): the Fortran-path raytracer with sub-boxes and
    global_pass, both from the oracle, zero heating tables as the reference passes them (evolve.py:193)."""
    import cases
    NumSrc = src_flux.shape[0]
    N = temp.shape[0]
    NumCells = N ** 3
    conv_criterion = min(int(conv * NumCells), (NumSrc - 1) / 3)
    prev1 = prev0 = 2 * NumCells
    xh_av = np.array(xh, dtype=np.float64, copy=True)
    xh_intermed = xh_av.copy()
    converged = False
    niter = 0
    phi = None
    while not converged and niter < max_iter:
        niter += 1
        phi = O.do_all_sources(src_flux, src_pos, max_subbox, subboxsize, sig, dr, ndens, xh_av, loss_fraction, thin,
                               thick, minlogtau, dlogtau, R)["phi_ion"]
        xh_av, xh_intermed, conv_flag, _ = O.global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi, cases.BH00,
                                                         cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
        s1 = np.sum(xh_intermed)
        s0 = np.sum(1.0 - xh_intermed)
        rel1 = abs((s1 - prev1) / s1) if s1 > 0 else 1.0
        rel0 = abs((s0 - prev0) / s0) if s0 > 0 else 1.0
        converged = (conv_flag < conv_criterion) or (rel1 < conv and rel0 < conv)
        prev1, prev0 = s1, s0
    return xh_intermed, phi, niter
