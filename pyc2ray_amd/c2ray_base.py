"""Thin C2Ray simulation class with the reference's interface (pyc2ray/c2ray_base.py).

Holds the parameters of a run, derives the constants the hot path needs, builds the radiation tables
and forwards ``evolve3D`` / ``do_raytracing`` to the MI355X kernels.  Same YAML layout, attribute names
and method names as the reference, so its driver scripts (e.g. test/paper_tests/test1_Ifront/run_test.py)
work with ``import pyc2ray_amd as pc2r``.

Differences from the reference:
  * ``do_raytracing`` passes the heating tables (the reference's method passes 16 arguments to a
    17-argument function, c2ray_base.py:314-321, and raises TypeError).
  * Cosmology: astropy's ``FlatLambdaCDM`` is used when astropy is installed; otherwise
    ``FlatLambdaCDMLite`` below evaluates the same flat LCDM expressions (matter + Lambda + photons +
    3.04 massless neutrinos).  The lite version could not be compared with astropy in the build
    container (astropy absent), so runs that need cosmological time steps to 1e-6 should have astropy.
  * ``use_mpi`` may be ``pyc2ray_amd.dist.MPI`` (torch.distributed/RCCL) as well as mpi4py's ``MPI``.
  * ``device_resident`` (default True; single GPU): ``ndens``, ``temp``, ``xh`` and ``phi_ion`` stay on the MI355X
    between time steps and cross PCIe only when the host touches them; see :class:`C2Ray`.
"""
import atexit
import re

import numpy as np
import yaml

try:
    from yaml import CSafeLoader as SafeLoader
except ImportError:  # pragma: no cover
    from yaml import SafeLoader

from .asora_core import cuda_is_init, device_close, device_init, photo_table_to_device
from . import _capi, _residency
from .evolve import evolve3D, evolve3D_MPI, evolve3D_resident
from .load_extensions import load_asora
from .radiation import BlackBodySource, make_tau_table
from .raytracing import do_raytracing
from .utils.logutils import printlog

__all__ = ['C2Ray', 'FlatLambdaCDMLite', 'YEAR', 'Mpc']

# Conversion factors with the values the reference hard-codes (c2ray_base.py:74-80)
pc = 3.086e18
YEAR = 3.15576E+07
ev2fr = 0.241838e15                     # eV to frequency (Hz)
ev2k = 1.0 / 8.617e-05                  # eV to Kelvin
kpc = 1e3 * pc
Mpc = 1e6 * pc
msun2g = 1.98892e33


class FlatLambdaCDMLite:
    """Flat LCDM background with the expressions astropy's FlatLambdaCDM(H0, Om0, Tcmb0, Ob0=...) uses
    (massless neutrinos, Neff = 3.04): E(z)^2 = Om0 (1+z)^3 + (Ogamma0 + Onu0)(1+z)^4 + Ode0.
    Only what C2Ray needs: age, lookback_time, scale_factor, and the inverse of age."""

    _G = 6.6743e-11               # CODATA 2018, SI
    _c = 299792458.0
    _sigma_sb = 5.670374419e-8
    _Mpc_m = 3.0856775814913673e22

    def __init__(self, H0, Om0, Tcmb0=0.0, Ob0=None, Neff=3.04):
        self.H0 = float(H0)
        self.Om0 = float(Om0)
        self.Ob0 = Ob0
        self.Tcmb0 = float(Tcmb0)
        self._H0_s = self.H0 * 1e3 / self._Mpc_m                       # s^-1
        rho_crit = 3.0 * self._H0_s ** 2 / (8.0 * np.pi * self._G)      # kg m^-3
        self.Ogamma0 = 4.0 * self._sigma_sb / self._c ** 3 * self.Tcmb0 ** 4 / rho_crit
        self.Onu0 = 0.22710731766 * Neff * self.Ogamma0
        self.Ode0 = 1.0 - self.Om0 - self.Ogamma0 - self.Onu0

    def efunc(self, z):
        zp1 = 1.0 + np.asarray(z, dtype=float)
        return np.sqrt(self.Om0 * zp1 ** 3 + (self.Ogamma0 + self.Onu0) * zp1 ** 4 + self.Ode0)

    def scale_factor(self, z):
        return 1.0 / (1.0 + z)

    def age(self, z):
        """Age of the universe at redshift z, in seconds."""
        from scipy.integrate import quad
        # t = 1/H0 int_0^{a} da' / (a' E(a'))  with a = 1/(1+z)
        a = 1.0 / (1.0 + z)
        f = lambda x: 1.0 / (x * float(self.efunc(1.0 / x - 1.0)))
        val, _ = quad(f, 0.0, a, epsabs=0.0, epsrel=1e-12, limit=200)
        return val / self._H0_s

    def lookback_time(self, z):
        return self.age(0.0) - self.age(z)

    def z_at_age(self, t):
        from scipy.optimize import brentq
        return brentq(lambda z: self.age(z) - t, -0.5, 1e4, xtol=1e-12, rtol=1e-12)


def _make_cosmology(H0, Om0, Tcmb0, Ob0):
    try:
        from astropy.cosmology import FlatLambdaCDM
        return FlatLambdaCDM(H0, Om0, Tcmb0, Ob0=Ob0), True
    except ImportError:
        return FlatLambdaCDMLite(H0, Om0, Tcmb0, Ob0=Ob0), False


class _DeviceGrid:
    """A grid attribute of C2Ray that may live on the device (see C2Ray.device_resident).  The host array is the
    attribute's value as always; what is tracked is which side holds the current data.  Reading the attribute hands
    out the host array, which the caller may then write into, so every read marks the device copy as out of date
    (and first fetches the data if the device holds the newer one); assigning replaces the host array."""

    def __init__(self, name, which):
        self.name, self.which = name, which

    def __set_name__(self, owner, attr):
        self.attr = "_grid_" + attr

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        try:
            arr = obj.__dict__[self.attr]
        except KeyError:
            raise AttributeError(self.name) from None
        if self.name in obj._device_newer:                   # results of the last step(s) still only on the device
            lib = load_asora()
            order = 'F' if (arr.flags.f_contiguous and not arr.flags.c_contiguous) else 'C'
            if self.name in ("xh", "phi_ion"):
                # into a FRESH array, as the reference binds a fresh array per step (c2ray_base.py:205-226): a caller that keeps
                # `prev = sim.xh` or appends sim.xh to a history must not see it change under its feet
                arr = lib.grid_to_host(self.which, lib.host_empty(arr.shape, order=order))
                obj.__dict__[self.attr] = arr
            else:
                # an input grid changed on the device (the density, diluted by cosmo_evolve): into the caller's OWN array, as the
                # reference scales it in place (c2ray_base.py:248) -- a reference kept to it sees the dilution
                if not (arr.flags.writeable and (arr.flags.c_contiguous or arr.flags.f_contiguous) and arr.dtype == np.float64):
                    arr = lib.host_empty(arr.shape, order=order)
                    obj.__dict__[self.attr] = arr
                lib.grid_to_host(self.which, arr)
                obj.__dict__.setdefault("_grid_fingerprints", {})[self.name] = obj._fingerprint(arr)
            obj._device_newer.discard(self.name)
        obj._host_newer.add(self.name)                       # the caller may modify what it gets
        return arr

    def __set__(self, obj, value):
        obj.__dict__[self.attr] = value
        obj._device_newer.discard(self.name)
        obj._host_newer.add(self.name)


class C2Ray:
    #: Default since round 6 (single GPU, use_gpu=True; set False on an instance for the reference's behaviour: every step uploads
    #: and downloads everything): ndens, temp, xh and phi_ion stay on the device between time steps.  evolve3D then uploads only
    #: the grids that were assigned or READ on the host since the last step (a read hands out the array, which may be written
    #: into), and downloads xh / phi_ion only when they are read; cosmo_evolve dilutes a density that lives on the device ON the
    #: device.  At 256^3 the five 128 MiB transfers of a time step cost as much as the step's outer iterations (33.2 against
    #: 17.4 ms per time step, profiles/r06_time_steps_resident.json).  xh and phi_ion behave as in the reference: every step
    #: binds a FRESH array (c2ray_base.py:205-226), an array kept from an earlier step keeps that step's values.  ndens and temp
    #: are the caller's arrays; writing into one through a reference kept from BEFORE the last step, without touching the
    #: attribute again (``n = sim.ndens`` ... evolve3D ... ``n *= 2``), is caught by a fingerprint of 16 384 samples of the host
    #: array taken at upload time (0.1 ms per grid and step at 256^3; any rescaling or whole-grid update changes it, an edit of a few cells may not: assign or read
    #: the attribute after such an edit, as ``sim.ndens[...] = v`` does by itself): the grid is uploaded again -- or, if the
    #: device copy has meanwhile been diluted by cosmo_evolve, so that the write went into stale values, a RuntimeError says so.
    device_resident = True
    _FINGERPRINT_SAMPLES = 16384

    ndens = _DeviceGrid("ndens", _capi.GRID_NDENS)
    temp = _DeviceGrid("temp", _capi.GRID_TEMP)
    xh = _DeviceGrid("xh", _capi.GRID_XH)
    phi_ion = _DeviceGrid("phi_ion", _capi.GRID_PHI_ION)

    def __init__(self, paramfile, Nmesh, use_gpu, use_mpi):
        """Basis class of a C2Ray simulation (pyc2ray/c2ray_base.py:82-145).

        paramfile : YAML parameter file (same keys as the reference's parameters.yml files)
        Nmesh     : mesh size
        use_gpu   : True = ASORA path, device-resident; False = the semantics of the reference's CPU raytracer
                    (sub-boxes, photon loss), evaluated on the GPU through the libc2ray-compatible entry points
        use_mpi   : None/False, mpi4py's MPI module, or pyc2ray_amd.dist.MPI
        """
        self._host_newer = set()           # grids whose host array holds newer data than the device
        self._device_newer = set()         # grids whose device copy holds newer data than the host array
        if use_mpi:
            self.mpi = use_mpi
            self.comm = use_mpi.COMM_WORLD
            self.rank = self.comm.Get_rank()
            self.nprocs = self.comm.Get_size()
        else:
            self.mpi = False
            self.rank = 0
            self.nprocs = 1

        self._read_paramfile(paramfile)
        self.N = Nmesh
        self.shape = (Nmesh, Nmesh, Nmesh)

        if use_gpu:
            self.gpu = True
            src_batch_size = self._ld["Raytracing"]["source_batch_size"]
            device_init(Nmesh, src_batch_size)
            atexit.register(self._gpu_close)
        else:
            self.gpu = False

        self._param_init()
        self._output_init()
        self._grid_init()
        self._cosmology_init()
        self._redshift_init()
        self._material_init()
        self._sources_init()
        self._radiation_init()
        if self.rank == 0:
            if self.gpu:
                q_max = np.ceil(1.73205080757 * min(self.R_max_LLS, 1.73205080757 * self.N / 2))
                self.printlog(f"Using ASORA Raytracing ( q_max = {q_max : n} )")
            else:
                self.printlog(f"Using CPU Raytracing (subboxsize = {self.subboxsize : n}, max_subbox = {self.max_subbox : n})")
            if self.mpi:
                self.printlog(f"Using {self.nprocs:n} MPI Ranks")
            else:
                self.printlog("Running in non-MPI (single-GPU/CPU) mode")
            self.printlog("Starting simulation... \n\n")

    # ---- time evolution -------------------------------------------------------------------------
    def set_timestep(self, z1, z2, num_timesteps):
        """Time step (s) between two redshift slices (c2ray_base.py:147-168)."""
        t1 = self._lookback_s(z1)
        t2 = self._lookback_s(z2)
        return (t1 - t2) / num_timesteps

    def evolve3D(self, dt, src_flux, src_pos):
        """Evolve the grid over one time step (c2ray_base.py:170-226)."""
        if self.device_resident and self.gpu and not self.mpi:
            return self._evolve3D_resident(dt, src_flux, src_pos)
        args = (self.temp, self.ndens, self.xh, self.photo_thin_table, self.photo_thick_table, self.minlogtau,
                self.dlogtau, self.R_max_LLS, self.convergence_fraction, self.sig, self.bh00, self.albpow,
                self.colh0, self.temph0, self.abu_c, self.logfile)
        if self.mpi and src_flux.shape[0] >= self.nprocs:
            self.xh, self.phi_ion = evolve3D_MPI(dt, self.dr, src_flux, src_pos, self.gpu, self.max_subbox,
                                                 self.subboxsize, self.loss_fraction, self.mpi, self.comm, self.rank,
                                                 self.nprocs, *args)
        else:
            self.xh, self.phi_ion = evolve3D(dt, self.dr, src_flux, src_pos, self.gpu, self.max_subbox,
                                             self.subboxsize, self.loss_fraction, *args)

    @classmethod
    def _fingerprint(cls, arr):
        """A strided sample of a host grid (a view of its memory in storage order): what the resident path compares to notice
        in-place changes made through a reference the caller kept."""
        flat = arr.ravel(order='K')
        step = max(1, flat.shape[0] // cls._FINGERPRINT_SAMPLES)
        return flat[::step].copy()

    def _written_behind_the_attribute(self, name):
        """Has the host array of input grid `name` changed since it was last in step with the device, without the attribute having
        been touched?  (``n = sim.ndens`` ... ``n *= 2``.)  True: the caller uploads it again.  If the DEVICE copy has moved on as
        well -- a density diluted on the device -- the two cannot be reconciled (the write went into stale values): that raises."""
        prints = self.__dict__.get("_grid_fingerprints", {})
        if name not in prints or name in self._host_newer:
            return False
        if np.array_equal(self._fingerprint(self.__dict__["_grid_" + name]), prints[name]):
            return False
        if name in self._device_newer:
            raise RuntimeError(f"C2Ray.{name} was written into through a reference kept from before the last time step, while the "
                               f"device held the newer copy (cosmo_evolve dilutes a device-resident density on the device).  Read "
                               f"sim.{name} again before modifying it (`sim.{name} *= f`, `sim.{name}[...] = v` do), or set "
                               f"sim.device_resident = False for the reference's all-through-the-host behaviour.")
        return True

    def _leave_device(self):
        """Someone else is about to overwrite the device grids (pyc2ray_amd/_residency.py): fetch what exists only there, and
        upload everything again at the next step."""
        for name in ("ndens", "temp", "xh", "phi_ion"):
            if name in self._device_newer and ("_grid_" + name) in self.__dict__:
                getattr(self, name)                              # (downloads; marks the host copy as the newer one)
        self._device_newer.clear()
        self._host_newer |= {"ndens", "temp", "xh"}
        self.__dict__.pop("_grid_fingerprints", None)

    def _evolve3D_resident(self, dt, src_flux, src_pos):
        """The same step with the grids left on the device (see `device_resident`)."""
        d = self.__dict__
        _residency.reclaim(except_for=self)                     # (another object's device copies, should there be one)
        _residency.register(self)
        uploads = {}
        prints = d.setdefault("_grid_fingerprints", {})
        for name, which in (("ndens", _capi.GRID_NDENS), ("temp", _capi.GRID_TEMP), ("xh", _capi.GRID_XH)):
            host = d["_grid_" + name]
            # (an input grid the host did not touch through the attribute may still have been written into through a kept reference)
            changed = name in self._host_newer or self._written_behind_the_attribute(name)
            if changed:
                uploads[which] = host
                if name != "xh":
                    prints[name] = self._fingerprint(host)
        phi_host = d.get("_grid_phi_ion")                       # (a subclass may never have assigned phi_ion)
        if phi_host is None or (phi_host.flags.f_contiguous and not phi_host.flags.c_contiguous):
            d["_grid_phi_ion"] = np.zeros(self.shape)           # the GPU path returns C-ordered rates (evolve.py:200)
        evolve3D_resident(dt, self.dr, src_flux, src_pos, uploads, self.N, self.photo_thin_table, self.minlogtau, self.dlogtau,
                          self.R_max_LLS, self.convergence_fraction, self.sig, self.bh00, self.albpow, self.colh0, self.temph0,
                          self.abu_c, self.logfile)
        self._host_newer -= {"ndens", "temp", "xh", "phi_ion"}
        self._device_newer |= {"xh", "phi_ion"}

    def cosmo_evolve(self, dt):
        """Advance time and redshift by dt, diluting density and rescaling the cell size when the run is
        cosmological; redshift is set at the half time step (c2ray_base.py:229-257)."""
        t_now = self.time
        t_half = t_now + 0.5 * dt
        t_after = t_now + dt
        z_half = self.time2zred(t_half)
        if self.cosmological:
            dilution_factor = ((1 + z_half) / (1 + self.zred)) ** 3
            if (self.device_resident and self.gpu and not self.mpi and cuda_is_init() and "ndens" not in self._host_newer
                    and "ndens" in self.__dict__.get("_grid_fingerprints", {}) and not self._written_behind_the_attribute("ndens")):
                # the density lives on the device and the host has not touched it since: diluted there (the same IEEE
                # multiplication per cell); the host array is fetched when someone reads sim.ndens
                load_asora().grid_scale(_capi.GRID_NDENS, dilution_factor)
                self._device_newer.add("ndens")
            else:
                self.ndens *= dilution_factor
            self.dr = self.dr_c * self._scale_factor(z_half)
        self.zred = z_half
        self.time = t_after

    def printlog(self, s, quiet=False):
        if self.logfile is None:
            raise RuntimeError("Please set the log file in output_ini")
        printlog(s, self.logfile, quiet)

    def write_output(self, z):
        pass

    # ---- utilities ------------------------------------------------------------------------------
    def time2zred(self, t):
        """Redshift at cosmic age t [s] (c2ray_base.py:281-284)."""
        if self._astropy:
            from astropy import units as u
            from astropy.cosmology import z_at_value
            return z_at_value(self.cosmology.age, t * u.s).value
        return self.cosmology.z_at_age(t)

    def zred2time(self, z, unit='s'):
        """Cosmic age at redshift z (c2ray_base.py:286-297)."""
        if self._astropy:
            return self.cosmology.age(z).to(unit).value
        t = self.cosmology.age(z)
        return {'s': t, 'yr': t / YEAR, 'Myr': t / (1e6 * YEAR), 'Gyr': t / (1e9 * YEAR)}[unit]

    def _lookback_s(self, z):
        if self._astropy:
            return self.cosmology.lookback_time(z).to('s').value
        return self.cosmology.lookback_time(z)

    def _scale_factor(self, z):
        return float(self.cosmology.scale_factor(z))

    def do_raytracing(self, src_flux, src_pos):
        """Photo-ionisation rates of the current state, chemistry untouched (c2ray_base.py:300-323)."""
        gamma = do_raytracing(self.dr, src_flux, src_pos, self.gpu, self.max_subbox, self.subboxsize,
                              self.loss_fraction, self.ndens, self.xh, self.photo_thin_table,
                              self.photo_thick_table, self.heat_thin_table, self.heat_thick_table, self.minlogtau,
                              self.dlogtau, self.R_max_LLS, self.sig, self.logfile)
        self.phi_ion = gamma[0]
        if gamma[1] is not None:
            self.phi_heat = gamma[1]
        return gamma

    # ---- initialisation (c2ray_base.py:329-480) ----------------------------------------------------
    def _param_init(self):
        """Constants of the run from the parameter file (c2ray_base.py:329-352)."""
        ld = self._ld
        for key in ('eth0', 'ethe0', 'ethe1', 'bh00', 'fh0', 'xih0', 'albpow'):
            setattr(self, key, ld['CGS'][key])
        for key in ('abu_h', 'abu_he', 'abu_c'):
            setattr(self, key, ld['Abundances'][key])
        for key in ('loss_fraction', 'convergence_fraction', 'max_subbox', 'subboxsize'):
            setattr(self, key, ld['Raytracing'][key])
        self.sig = ld['Photo']['sigma_HI_at_ion_freq']
        self.mean_molecular = self.abu_h + 4.0 * self.abu_he
        # collisional ionisation parameter and ionisation energy in K (c2ray_base.py:346-347)
        self.colh0 = ld['CGS']['colh0_fact'] * self.fh0 * self.xih0 / self.eth0 ** 2
        self.temph0 = self.eth0 * ev2k

    def _cosmology_init(self):
        cs = self._ld['Cosmology']
        self.cosmology, self._astropy = _make_cosmology(100 * cs['h'], cs['Omega0'], cs['cmbtemp'], cs['Omega_B'])
        self.cosmological = cs['cosmological']
        self.zred_0 = cs['zred_0']
        self.age_0 = self.zred2time(self.zred_0)
        if self.cosmological:
            if self.rank == 0:
                self.printlog(f"Cosmology is on, scaling comoving quantities to the initial redshift, which is z0 = {self.zred_0:.3f}...")
            self.dr = self._scale_factor(self.zred_0) * self.dr_c
        else:
            if self.rank == 0:
                self.printlog("Cosmology is off.")

    def _radiation_init(self):
        ph = self._ld['Photo']
        self.minlogtau = ph['minlogtau']
        self.maxlogtau = ph['maxlogtau']
        self.NumTau = ph['NumTau']
        self.SourceType = ph['SourceType']
        self.grey = ph['grey']
        self.compute_heating_rates = ph['compute_heating_rates']
        if self.rank == 0:
            if self.grey:
                self.printlog("Warning: Using grey opacity")
            else:
                self.printlog(f"Using power-law opacity with {self.NumTau:n} table points between tau=10^({self.minlogtau:n}) and tau=10^({self.maxlogtau:n})")
        # NumTau + 1 points: tau = 0 first, then log-spaced (as in C2Ray)
        self.tau, self.dlogtau = make_tau_table(self.minlogtau, self.maxlogtau, self.NumTau)
        ion_freq_HI = ev2fr * self.eth0
        ion_freq_HeII = ev2fr * self.ethe1
        if self.SourceType == 'blackbody':
            freq_min = ion_freq_HI
            freq_max = 10 * ion_freq_HeII
            self.bb_Teff = self._ld['BlackBodySource']['Teff']
            self.cs_pl_idx_h = self._ld['BlackBodySource']['cross_section_pl_index']
            radsource = BlackBodySource(self.bb_Teff, self.grey, ion_freq_HI, self.cs_pl_idx_h)
            if self.rank == 0:
                self.printlog(f"Using Black-Body sources with effective temperature T = {radsource.temp :.1e} K")
                self.printlog(f"Spectrum Frequency Range: {freq_min:.3e} to {freq_max:.3e} Hz")
                self.printlog("Integrating photoionization rates tables...")
            self.photo_thin_table, self.photo_thick_table = radsource.make_photo_table(self.tau, freq_min, freq_max, 1e48)
            if self.compute_heating_rates:
                self.printlog("Integrating photoheating rates tables...")
                self.heat_thin_table, self.heat_thick_table = radsource.make_heat_table(self.tau, freq_min, freq_max, 1e48)
            else:
                self.printlog("INFO: No heating rates")
                self.heat_thin_table = np.zeros(self.NumTau + 1)
                self.heat_thick_table = np.zeros(self.NumTau + 1)
        else:
            raise NameError("Unknown source type : ", self.SourceType)
        if self.gpu:
            photo_table_to_device(self.photo_thin_table, self.photo_thick_table)
            if self.rank == 0:
                self.printlog("Successfully copied radiation tables to GPU memory.")

    def _grid_init(self):
        self.boxsize_c = self._ld['Grid']['boxsize'] * Mpc
        self.dr_c = self.boxsize_c / self.N
        if self.rank == 0:
            self.printlog(f"Welcome! Mesh size is N = {self.N:n}.")
            self.printlog(f"Simulation Box size (comoving Mpc): {self.boxsize_c/Mpc:.3e}")
        self.dr = self.dr_c
        # R_max (LLS type 3) in cell units
        self.R_max_LLS = self._ld['Photo']['R_max_cMpc'] * self.N / self._ld['Grid']['boxsize']
        self.printlog(f"Maximum comoving distance for photons from source (type 3 LLS): {self._ld['Photo']['R_max_cMpc'] : .3e} comoving Mpc")
        self.printlog(f"This corresponds to                                             {self.R_max_LLS : .3f} grid cells.")

    # overridden by the concrete simulation classes
    def _output_init(self):
        pass

    def _redshift_init(self):
        pass

    def _material_init(self):
        pass

    def _sources_init(self):
        pass

    def _read_paramfile(self, paramfile):
        """YAML reader that also takes 1e4-style numbers as floats (c2ray_base.py:490-507)."""
        class _Loader(SafeLoader):
            pass
        _Loader.add_implicit_resolver(
            u'tag:yaml.org,2002:float',
            re.compile(u'''^(?:
            [-+]?(?:[0-9][0-9_]*)\\.[0-9_]*(?:[eE][-+]?[0-9]+)?
            |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
            |\\.[0-9_]+(?:[eE][-+][0-9]+)?
            |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\\.[0-9_]*
            |[-+]?\\.(?:inf|Inf|INF)
            |\\.(?:nan|NaN|NAN))$''', re.X),
            list(u'-+0123456789.'))
        with open(paramfile, 'r') as f:
            self._ld = yaml.load(f, _Loader)

    def _gpu_close(self):
        if cuda_is_init():
            device_close()
