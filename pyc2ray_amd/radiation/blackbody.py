"""Black-body photo-ionisation tables, as pyc2ray/radiation/blackbody.py:20-89.

Produces the two tables the raytracing kernel interpolates (rates.cu:16-41):
    thick(tau) = int L_nu/(h nu) exp(-tau a(nu)) dnu,   thin(tau) = int L_nu/(h nu) a(nu) exp(-tau a(nu)) dnu
with a(nu) = (nu/nu0)^-p (or 1 when grey), normalised so that thick(0) = S_star_ref.
Physical constants carry the exact values the reference hard-codes for consistency with the
original C2-Ray (blackbody.py:10-13,17); the heating tables need h and the Rydberg frequency,
which the reference takes from astropy and which are spelled out here (CODATA 2018).
"""
import numpy as np
from scipy.integrate import quad, quad_vec

__all__ = ['BlackBodySource']

h_over_k = 6.6260755e-27 / 1.381e-16          # blackbody.py:10
pi = 3.141592654                              # blackbody.py:11
c = 2.997925e+10                              # blackbody.py:12
two_pi_over_c_square = 2.0 * pi / (c * c)     # blackbody.py:13
hplanck = 6.62607015e-27                      # astropy.constants.h in cgs   (blackbody.py:14)
ion_freq_HI = 3.2898419602508e15              # astropy (Ryd*c) in Hz        (blackbody.py:15)
sigma_0 = 6.3e-18


class BlackBodySource:
    """A point source with a black-body spectrum of temperature `temp` [K]."""

    def __init__(self, temp, grey, freq0, pl_index) -> None:
        self.temp = temp
        self.grey = grey
        self.freq0 = freq0
        self.pl_index = pl_index
        self.R_star = 1.0

    def SED(self, freq):
        if freq * h_over_k / self.temp < 700.0:
            return 4 * np.pi * self.R_star ** 2 * two_pi_over_c_square * freq ** 2 / (np.exp(freq * h_over_k / self.temp) - 1.0)
        return 0.0

    def integrate_SED(self, f1, f2):
        return quad(self.SED, f1, f2)[0]

    def normalize_SED(self, f1, f2, S_star_ref):
        S_unscaled = self.integrate_SED(f1, f2)
        self.R_star = np.sqrt(S_star_ref / S_unscaled) * self.R_star

    def cross_section_freq_dependence(self, freq):
        if self.grey:
            return 1.0
        return (freq / self.freq0) ** (-self.pl_index)

    def _photo_thick_integrand_vec(self, freq, tau):
        a = self.cross_section_freq_dependence(freq)
        with np.errstate(over='ignore', under='ignore'):
            itg = self.SED(freq) * np.exp(-tau * a)
        return np.where(tau * a < 700.0, itg, 0.0)

    def _photo_thin_integrand_vec(self, freq, tau):
        a = self.cross_section_freq_dependence(freq)
        with np.errstate(over='ignore', under='ignore'):
            itg = self.SED(freq) * a * np.exp(-tau * a)
        return np.where(tau * a < 700.0, itg, 0.0)

    def _heat_thick_integrand_vec(self, freq, tau):
        return hplanck * (freq - ion_freq_HI) * self._photo_thick_integrand_vec(freq, tau)

    def _heat_thin_integrand_vec(self, freq, tau):
        return hplanck * (freq - ion_freq_HI) * self._photo_thin_integrand_vec(freq, tau)

    def make_photo_table(self, tau, freq_min, freq_max, S_star_ref):
        """(table_thin, table_thick) on the optical depths `tau` (blackbody.py:71-77)."""
        self.normalize_SED(freq_min, freq_max, S_star_ref)
        table_thin = quad_vec(lambda f: self._photo_thin_integrand_vec(f, tau), freq_min, freq_max, epsrel=1e-12)[0]
        table_thick = quad_vec(lambda f: self._photo_thick_integrand_vec(f, tau), freq_min, freq_max, epsrel=1e-12)[0]
        return table_thin, table_thick

    def make_heat_table(self, tau, freq_min, freq_max, S_star_ref):
        """(heat_thin, heat_thick) (blackbody.py:79-85)."""
        self.normalize_SED(freq_min, freq_max, S_star_ref)
        table_thin = quad_vec(lambda f: self._heat_thin_integrand_vec(f, tau), freq_min, freq_max, epsrel=1e-12)[0]
        table_thick = quad_vec(lambda f: self._heat_thick_integrand_vec(f, tau), freq_min, freq_max, epsrel=1e-12)[0]
        return table_thin, table_thick
