"""C2Ray_Test: test-case simulations with constant density and simple source files, with the
interface of pyc2ray/c2ray_test.py."""
import pickle as pkl

import numpy as np

from .c2ray_base import C2Ray, YEAR
from .utils.sourceutils import read_test_sources

__all__ = ['C2Ray_Test']


class C2Ray_Test(C2Ray):
    def __init__(self, paramfile, Nmesh, use_gpu, use_mpi=None):
        """A C2Ray test-case simulation (c2ray_test.py:14-29)."""
        super().__init__(paramfile, Nmesh, use_gpu, use_mpi)
        if self.rank == 0:
            self.printlog('Running: "C2Ray Test"')

    def read_sources(self, file, numsrc, S_star_ref=1e48):
        """Read `numsrc` sources of a Test-C2Ray source file: returns (src_pos (3,numsrc) 1-based,
        src_flux normalised by S_star_ref) (c2ray_test.py:31-62)."""
        return read_test_sources(file, numsrc, S_star_ref)

    def density_init(self, z):
        """Constant density from the parameter file, scaled to redshift (c2ray_test.py:65-77)."""
        self.set_constant_average_density(self.avg_dens, z)

    def _write_pickles(self, suffix):
        for stem, grid in (("xfrac", self.xh), ("IonRates", self.phi_ion)):
            with open(f"{self.results_basename}{stem}{suffix}", "wb") as f:
                pkl.dump(grid, f)

    def write_output(self, z):
        """Ionised fraction and rates as `xfrac_<z>.pkl` / `IonRates_<z>.pkl` (c2ray_test.py:79-91)."""
        self._write_pickles(f"_{z:.3f}.pkl")

    def write_output_numbered(self, n):
        """Same, numbered instead of named by redshift (c2ray_test.py:93-105)."""
        self._write_pickles(f"_{n:n}.pkl")

    def set_constant_average_density(self, ndens, z):
        """Density grid = ndens (comoving, i.e. proper at z = 0) scaled by (1+z)^3; when the run is not
        cosmological the initial redshift of the parameter file is used (c2ray_test.py:107-126)."""
        redshift = z if self.cosmological else self.zred_0
        self.ndens = ndens * np.ones(self.shape, order='F') * (1 + redshift) ** 3

    def generate_redshift_array(self, num_zred, delta_t):
        """num_zred redshifts separated by delta_t years of cosmic time, starting at the initial redshift
        (c2ray_test.py:128-152)."""
        ages = self.age_0 + np.arange(num_zred) * (delta_t * YEAR)
        return np.array([self.time2zred(t) for t in ages])

    # ---- overridden initialisation (c2ray_test.py:158-181) ------------------------------------------
    def _redshift_init(self):
        self.time = self.age_0
        self.zred = self.zred_0

    def _material_init(self):
        mat = self._ld['Material']
        self.avg_dens = mat['avg_dens']
        self.ndens = np.empty(self.shape, order='F')           # filled by density_init
        self.xh = np.full(self.shape, float(mat['xh0']), order='F')
        self.temp = np.full(self.shape, float(mat['temp0']), order='F')
        self.phi_ion = np.zeros(self.shape, order='F')

    def _output_init(self):
        self.results_basename = self._ld['Output']['results_basename']
        self.logfile = self.results_basename + self._ld['Output']['logfile']
        if self.rank == 0:
            with open(self.logfile, "w") as f:
                f.write("\nLog file for pyC2Ray \n\n")
