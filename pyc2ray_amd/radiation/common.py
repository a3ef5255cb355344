"""Optical-depth table, as pyc2ray/radiation/common.py:13-37."""
import numpy as np

__all__ = ['make_tau_table']


def make_tau_table(minlogtau, maxlogtau, NumTau):
    """tau[0] = 0, tau[1:] = 10**(minlogtau + arange(NumTau)*dlogtau); returns (tau, dlogtau).
    The table has NumTau+1 entries (same convention as C2-Ray)."""
    dlogtau = (maxlogtau - minlogtau) / NumTau
    tau = np.empty(NumTau + 1)
    tau[0] = 0.0
    tau[1:] = 10 ** (minlogtau + np.arange(NumTau) * dlogtau)
    return tau, dlogtau
