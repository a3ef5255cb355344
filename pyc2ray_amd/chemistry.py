"""Chemistry-only convenience wrapper, as pyc2ray/chemistry.py:43-95."""
import numpy as np

from .load_extensions import load_c2ray

__all__ = ['hydrogenODE']

#: (13.598 eV / k_B) in K -- what the reference derives with astropy (chemistry.py:88)
TEMPH0_K = 13.598 * 11604.518121550082


def hydrogenODE(dt, ndens, temp, xh, phi_ion, bh00=2.59e-13, albpow=-0.7, colh0=1.3e-8, abu_c=7.1e-7):
    """Advance the hydrogen ionised fraction of every cell by `dt` seconds at fixed Gamma.

    Same signature and defaults as the reference.  The reference passes one array as xh, xh_av AND
    xh_intermed to the Fortran (chemistry.py:84,91) -- aliased intent(inout) dummies, whose
    surviving store is compiler-dependent; the value its tutorial prints (<x> = 0.127,
    tutorials/chemistry_solver.ipynb cell 5) is the end-of-step fraction, which is what is returned
    here.  Raises AssertionError when 1% or more of the cells fail the convergence test, like the
    reference (chemistry.py:93-94)."""
    shape = np.shape(xh)
    xh0 = np.array(xh, dtype=np.float64, order='F', copy=True)
    xh_av = xh0.copy(order='F')
    xh_intermed = xh0.copy(order='F')
    b = lambda a: np.asfortranarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))
    conv_flag = load_c2ray().chemistry.global_pass(dt, b(ndens), b(temp), xh0, xh_av, xh_intermed, b(phi_ion),
                                                   bh00, albpow, colh0, TEMPH0_K, abu_c)
    convergence = conv_flag / np.size(xh_intermed)
    assert convergence < 0.01
    return xh_intermed
