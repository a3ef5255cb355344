"""Black-body radiation tables (the role of pyc2ray/radiation/blackbody.py:20-89).

For a source with photon spectrum S(nu) (black body of temperature T, normalised to S_star_ref photons/s between
the integration limits) and a cross section sigma(nu) = sigma_0 a(nu), a(nu) = (nu/nu0)^-p (a = 1 when grey), the
raytracing kernel interpolates (src/asora/rates.cu:16-41)

    thick(tau) = int S(nu) exp(-tau a(nu)) dnu          thin(tau) = int S(nu) a(nu) exp(-tau a(nu)) dnu

and the heating counterparts carry the extra factor h (nu - nu_HI).  All four are the same integral with two
switches, which is how they are written here.  The physical constants are the rounded values the reference
hard-codes for consistency with the original C2-Ray (blackbody.py:10-13); h and the Rydberg frequency, which the
reference takes from astropy, are spelled out (CODATA 2018).
"""
import numpy as np
from scipy.integrate import quad, quad_vec

__all__ = ['BlackBodySource']

h_over_k = 6.6260755e-27 / 1.381e-16
pi = 3.141592654
c = 2.997925e+10
two_pi_over_c_square = 2.0 * pi / (c * c)
hplanck = 6.62607015e-27                      # erg s
ion_freq_HI = 3.2898419602508e15              # Hz
sigma_0 = 6.3e-18

_EXP_CUTOFF = 700.0                           # exponents beyond this are treated as "no photons" (blackbody.py:32,50)


class BlackBodySource:
    """Point source with a black-body spectrum; same constructor and methods as the reference's class."""

    def __init__(self, temp, grey, freq0, pl_index) -> None:
        self.temp, self.grey, self.freq0, self.pl_index = temp, grey, freq0, pl_index
        self.R_star = 1.0

    # ---- spectrum ---------------------------------------------------------------------------------------
    def SED(self, freq):
        """Photons per second per Hz of a star of radius R_star: 4 pi R^2 x (2 pi/c^2) nu^2 / (exp(h nu/kT) - 1)."""
        x = freq * h_over_k / self.temp
        if x >= _EXP_CUTOFF:
            return 0.0
        surface = 4 * np.pi * self.R_star ** 2
        return surface * two_pi_over_c_square * freq ** 2 / (np.exp(x) - 1.0)

    def integrate_SED(self, f1, f2):
        return quad(self.SED, f1, f2)[0]

    def normalize_SED(self, f1, f2, S_star_ref):
        """Scale R_star so that the photon output between f1 and f2 is S_star_ref."""
        self.R_star *= np.sqrt(S_star_ref / self.integrate_SED(f1, f2))

    def cross_section_freq_dependence(self, freq):
        return 1.0 if self.grey else (freq / self.freq0) ** (-self.pl_index)

    # ---- tables -----------------------------------------------------------------------------------------
    def _integrand(self, freq, tau, thin, heating):
        a = self.cross_section_freq_dependence(freq)
        depth = tau * a
        with np.errstate(over='ignore', under='ignore'):
            value = self.SED(freq) * np.exp(-depth)
        value = np.where(depth < _EXP_CUTOFF, value, 0.0)
        if thin:
            value = value * a
        if heating:
            value = value * (hplanck * (freq - ion_freq_HI))
        return value

    def _table_pair(self, tau, freq_min, freq_max, S_star_ref, heating):
        self.normalize_SED(freq_min, freq_max, S_star_ref)
        tau = np.asarray(tau, dtype=float)
        integrate = lambda thin: quad_vec(lambda f: self._integrand(f, tau, thin, heating), freq_min, freq_max,
                                          epsrel=1e-12)[0]
        return integrate(True), integrate(False)

    def make_photo_table(self, tau, freq_min, freq_max, S_star_ref):
        """(thin, thick) photo-ionisation tables on the optical depths `tau`."""
        return self._table_pair(tau, freq_min, freq_max, S_star_ref, heating=False)

    def make_heat_table(self, tau, freq_min, freq_max, S_star_ref):
        """(thin, thick) photo-heating tables on the optical depths `tau`."""
        return self._table_pair(tau, freq_min, freq_max, S_star_ref, heating=True)
