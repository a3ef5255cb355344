"""Extension loader with the reference's names (pyc2ray/load_extensions.py:9-48).

The reference imports two CPython extension modules, ``pyc2ray.lib.libasora`` (CUDA) and
``pyc2ray.lib.libc2ray`` (f2py Fortran).  Here both are thin objects over ONE ctypes-loaded
shared library, ``pyc2ray_amd/lib/libasora_hip.so``; their methods keep the reference's names,
argument order and in-place conventions, so code written against the reference's extension
modules (pyc2ray/evolve.py:147-210, raytracing_benchmark/run_test.py:66-85) runs unchanged.

Differences, all deliberate:
  * a missing library raises RuntimeError for BOTH loaders (the reference makes ASORA optional and
    prints an "Info" line, load_extensions.py:41-44).  This build has no CPU path to fall back to:
    ``libc2ray.raytracing.do_all_sources`` and ``libc2ray.chemistry.global_pass`` keep the semantics of
    the reference's CPU functions but run on the GPU.
  * errors inside the library come back as RuntimeError carrying the library's message instead of
    an uncaught C++ exception.
  * array arguments are validated (dtype float64 / int32, contiguity, size) instead of being
    trusted (src/asora/python_module.cu:60-63).
"""
import ctypes as C

import numpy as np

from . import _capi, _pinned

__all__ = ["load_c2ray", "load_asora"]


def _flat_f64(a, n, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError(f"{name} must be Array of type double")       # python_module.cu:55
    if not (a.flags.c_contiguous or a.flags.f_contiguous):
        raise ValueError(f"{name} must be contiguous")
    if a.size != n:
        raise ValueError(f"{name} has {a.size} elements, expected {n}")
    return a


class _LibAsora:
    """Stand-in for the reference's ``libasora`` module (src/asora/python_module.cu:153-161)."""

    def __init__(self, lib):
        self._lib = lib
        self._N = None

    # ---- the six reference methods --------------------------------------------------------
    def device_init(self, N, num_src_par, device_id=None):
        # PYC2RAY_AMD_OPTIONS="13=2,14=2": library options (asora_set_option; the numbers are those of include/asora_hip.h) applied
        # around every device_init -- before it for the ones device_init itself reads (17: placement candidates), after it as
        # well.  For running an existing script or the test suite with a non-default variant.
        import os

        def apply_env_options():
            for item in filter(None, os.environ.get("PYC2RAY_AMD_OPTIONS", "").split(",")):
                opt, _, val = item.partition("=")
                self.set_option(int(opt), int(val))

        apply_env_options()
        if device_id is None:
            _capi.check(self._lib.asora_device_init(int(N), int(num_src_par)), "device_init")
        else:
            _capi.check(self._lib.asora_device_init_ex(int(N), int(num_src_par), int(device_id)), "device_init")
        self._N = int(N)
        apply_env_options()

    def device_close(self):
        _capi.check(self._lib.asora_device_close(), "device_close")
        self._N = None

    def density_to_device(self, ndens, N):
        a = _flat_f64(ndens, int(N) ** 3, "ndens")
        _capi.check(self._lib.asora_density_to_device(_capi.dptr(a), int(N)), "density_to_device")

    def photo_table_to_device(self, thin_table, thick_table, NumTau):
        t0 = np.ascontiguousarray(thin_table, dtype=np.float64)
        t1 = np.ascontiguousarray(thick_table, dtype=np.float64)
        if t0.size < NumTau or t1.size < NumTau:
            raise ValueError("photo_table_to_device: NumTau exceeds the table length")
        _capi.check(self._lib.asora_photo_table_to_device(_capi.dptr(t0), _capi.dptr(t1), int(NumTau)),
                    "photo_table_to_device")

    def heat_table_to_device(self, heat_thin_table, heat_thick_table, NumTau):
        """Extension: photo-heating tables (the reference's GPU library has none)."""
        t0 = np.ascontiguousarray(heat_thin_table, dtype=np.float64)
        t1 = np.ascontiguousarray(heat_thick_table, dtype=np.float64)
        _capi.check(self._lib.asora_heat_table_to_device(_capi.dptr(t0), _capi.dptr(t1), int(NumTau)),
                    "heat_table_to_device")

    def source_data_to_device(self, pos, flux, NumSrc):
        p = np.ascontiguousarray(pos)
        if p.dtype != np.int32:
            raise TypeError("source positions must be int32 (use format_sources)")
        f = np.ascontiguousarray(flux, dtype=np.float64)
        if p.size < 3 * NumSrc or f.size < NumSrc:
            raise ValueError("source_data_to_device: arrays shorter than NumSrc")
        _capi.check(self._lib.asora_source_data_to_device(_capi.iptr(p), _capi.dptr(f), int(NumSrc)),
                    "source_data_to_device")

    def do_all_sources(self, R, coldensh_out, sig, dr, ndens, xh_av, phi_ion, NumSrc, m1,
                       minlogtau, dlogtau, NumTau):
        n = int(m1) ** 3
        if not isinstance(coldensh_out, np.ndarray) or coldensh_out.dtype != np.float64:
            raise TypeError("coldensh_out must be Array of type double")   # python_module.cu:53-57
        x = _flat_f64(xh_av, n, "xh_av")
        ph = _flat_f64(phi_ion, n, "phi_ion")
        if not ph.flags.writeable:
            raise ValueError("phi_ion must be writeable (it is filled in place)")
        _capi.check(self._lib.asora_do_all_sources(float(R), None, float(sig), float(dr), None, _capi.dptr(x),
                                                   _capi.dptr(ph), int(NumSrc), int(m1), float(minlogtau),
                                                   float(dlogtau), int(NumTau)), "do_all_sources")
        return None

    # ---- device-resident extension (include/asora_hip.h section B/C) ------------------------
    def grid_to_device(self, which, a):
        N = a.shape[0]
        if a.ndim != 3 or a.shape != (N, N, N):
            raise ValueError("grid must have shape (N,N,N)")
        a = np.asarray(a, dtype=np.float64)
        if a.flags.c_contiguous:
            order = b'C'
        elif a.flags.f_contiguous:
            order = b'F'
        else:
            a, order = np.ascontiguousarray(a), b'C'
        _capi.check(self._lib.asora_grid_to_device(int(which), _capi.dptr(a), int(N), order), "grid_to_device")

    def grid_to_host(self, which, out):
        N = out.shape[0]
        if out.dtype != np.float64 or out.shape != (N, N, N):
            raise ValueError("output grid must be float64 of shape (N,N,N)")
        if out.flags.c_contiguous:
            _capi.check(self._lib.asora_grid_to_host(int(which), _capi.dptr(out), int(N), b'C'), "grid_to_host")
        elif out.flags.f_contiguous:
            _capi.check(self._lib.asora_grid_to_host(int(which), _capi.dptr(out), int(N), b'F'), "grid_to_host")
        else:
            tmp = np.empty((N, N, N))
            _capi.check(self._lib.asora_grid_to_host(int(which), _capi.dptr(tmp), int(N), b'C'), "grid_to_host")
            out[...] = tmp
        return out

    def grid_copy(self, dst, src):
        _capi.check(self._lib.asora_grid_copy(int(dst), int(src)), "grid_copy")

    def grid_scale(self, which, factor):
        """grid *= factor on the device (c2ray_base.py:248 for a device-resident density)."""
        _capi.check(self._lib.asora_grid_scale(int(which), float(factor)), "grid_scale")

    def grid_sum(self, which):
        out = C.c_double(0.0)
        _capi.check(self._lib.asora_grid_sum(int(which), C.byref(out)), "grid_sum")
        return out.value

    def host_empty(self, shape, order='C'):
        """An uninitialised float64 array for a grid download, in page-locked memory when there is some (_pinned.py)."""
        return _pinned.empty(self._lib, shape, order)

    def device_ptr(self, which):
        return self._lib.asora_device_ptr(int(which))

    def raytrace_device(self, R, sig, dr, src_begin, src_count, minlogtau, dlogtau, NumTau):
        _capi.check(self._lib.asora_raytrace_device(float(R), float(sig), float(dr), int(src_begin), int(src_count),
                                                    float(minlogtau), float(dlogtau), int(NumTau)),
                    "raytrace_device")

    def raytrace_begin(self, R, sig, dr, minlogtau, dlogtau, NumTau):
        _capi.check(self._lib.asora_raytrace_begin(float(R), float(sig), float(dr), float(minlogtau), float(dlogtau),
                                                   int(NumTau)), "raytrace_begin")

    def raytrace_begin_planes(self, R, sig, dr, minlogtau, dlogtau, NumTau, runs):
        """runs: [(first plane, number of planes), ...] -- see asora_raytrace_begin_planes."""
        r = np.ascontiguousarray(np.asarray(runs, dtype=np.int32).reshape(-1, 2))
        _capi.check(self._lib.asora_raytrace_begin_planes(float(R), float(sig), float(dr), float(minlogtau),
                                                          float(dlogtau), int(NumTau), _capi.iptr(r), int(r.shape[0])),
                    "raytrace_begin_planes")

    def raytrace_range(self, src_begin, src_count):
        _capi.check(self._lib.asora_raytrace_range(int(src_begin), int(src_count)), "raytrace_range")

    def raytrace_fold(self, i_begin, i_count):
        _capi.check(self._lib.asora_raytrace_fold(int(i_begin), int(i_count)), "raytrace_fold")

    def stream_ptr(self):
        return self._lib.asora_stream()

    def chemistry_device(self, dt, bh00, albpow, colh0, temph0, abu_c):
        conv = C.c_int(0)
        s1 = C.c_double(0.0)
        s0 = C.c_double(0.0)
        _capi.check(self._lib.asora_chemistry_device(float(dt), float(bh00), float(albpow), float(colh0),
                                                     float(temph0), float(abu_c), C.byref(conv), C.byref(s1),
                                                     C.byref(s0)), "chemistry_device")
        return conv.value, s1.value, s0.value

    def device_init_auto(self, N):
        _capi.check(self._lib.asora_device_init_auto(int(N)), "device_init_auto")

    def subbox_raytrace_device(self, max_subbox, subboxsize, loss_fraction, R, sig, dr, minlogtau, dlogtau, NumTau,
                               src_begin, src_count):
        """(sum_nbox, photon_loss): the sub-box raytracer on the device-resident grids and uploaded sources."""
        nbox = C.c_int(0)
        loss = C.c_double(0.0)
        _capi.check(self._lib.asora_subbox_raytrace_device(int(max_subbox), int(subboxsize), float(loss_fraction), float(R),
                                                           float(sig), float(dr), float(minlogtau), float(dlogtau),
                                                           int(NumTau), int(src_begin), int(src_count), C.byref(nbox),
                                                           C.byref(loss)), "subbox_raytrace_device")
        return nbox.value, loss.value

    def chemistry_range(self, dt, bh00, albpow, colh0, temph0, abu_c, i_begin, i_count, first):
        _capi.check(self._lib.asora_chemistry_range(float(dt), float(bh00), float(albpow), float(colh0), float(temph0),
                                                    float(abu_c), int(i_begin), int(i_count), int(bool(first))),
                    "chemistry_range")

    def chemistry_finish(self):
        conv = C.c_int(0)
        s1 = C.c_double(0.0)
        s0 = C.c_double(0.0)
        _capi.check(self._lib.asora_chemistry_finish(C.byref(conv), C.byref(s1), C.byref(s0)), "chemistry_finish")
        return conv.value, s1.value, s0.value

    def reduction_ptr(self):
        """Device address of {sum x, sum 1-x, conv_flag} of the chemistry_range calls so far (three doubles)."""
        return self._lib.asora_reduction_ptr()

    def evolve_begin(self, dt, bh00, albpow, colh0, temph0, abu_c, R, sig, dr, minlogtau, dlogtau, NumTau,
                     src_begin, src_count, conv_criterion, convergence_fraction):
        """Start a time step of the device-resident loop (NDENS, TEMP, XH on the device)."""
        _capi.check(self._lib.asora_evolve_begin(float(dt), float(bh00), float(albpow), float(colh0), float(temph0),
                                                 float(abu_c), float(R), float(sig), float(dr), float(minlogtau),
                                                 float(dlogtau), int(NumTau), int(src_begin), int(src_count),
                                                 float(conv_criterion), float(convergence_fraction)), "evolve_begin")

    def evolve_enqueue(self, iterations):
        _capi.check(self._lib.asora_evolve_enqueue(int(iterations)), "evolve_enqueue")

    def evolve_poll(self, max_rows=32):
        """(niter, converged, rows): rows[q] = (conv_flag, sum_xh1, sum_xh0, rel_change_xh1, rel_change_xh0) of the
        iterations carried out since the last poll."""
        niter, done, got = C.c_int(0), C.c_int(0), C.c_int(0)
        hist = np.zeros((int(max_rows), 5))
        # max_rows = 0: the caller gives the rows up (the library's history ring holds 64 iterations between polls)
        _capi.check(self._lib.asora_evolve_poll(C.byref(niter), C.byref(done), _capi.dptr(hist) if max_rows > 0 else None,
                                                int(max_rows), C.byref(got)), "evolve_poll")
        return niter.value, bool(done.value), hist[:got.value]

    # ---- the device-resident loop with the sources sharded over several GPUs (asora_evolve_slab_*) ----
    def evolve_begin_slab(self, dt, bh00, albpow, colh0, temph0, abu_c, R, sig, dr, minlogtau, dlogtau, NumTau,
                          src_begin, src_count, conv_criterion, convergence_fraction, own_begin, own_count):
        _capi.check(self._lib.asora_evolve_begin_slab(float(dt), float(bh00), float(albpow), float(colh0), float(temph0),
                                                      float(abu_c), float(R), float(sig), float(dr), float(minlogtau),
                                                      float(dlogtau), int(NumTau), int(src_begin), int(src_count),
                                                      float(conv_criterion), float(convergence_fraction), int(own_begin),
                                                      int(own_count)), "evolve_begin_slab")

    def evolve_slab_trace(self, src_begin, src_count):
        _capi.check(self._lib.asora_evolve_slab_trace(int(src_begin), int(src_count)), "evolve_slab_trace")

    def evolve_slab_fold_out(self, i_begin, i_count):
        _capi.check(self._lib.asora_evolve_slab_fold_out(int(i_begin), int(i_count)), "evolve_slab_fold_out")

    def evolve_slab_outbox_ptr(self):
        return self._lib.asora_evolve_slab_outbox()

    def evolve_slab_outbox_to_host(self, i_begin, i_count, N):
        out = np.empty((int(i_count), N, N))
        _capi.check(self._lib.asora_evolve_slab_outbox_to_host(int(i_begin), int(i_count), _capi.dptr(out)), "evolve_slab_outbox_to_host")
        return out

    def evolve_slab_outbox_from_host(self, i_begin, planes):
        a = np.ascontiguousarray(planes, dtype=np.float64)
        _capi.check(self._lib.asora_evolve_slab_outbox_from_host(int(i_begin), int(a.shape[0]), _capi.dptr(a)), "evolve_slab_outbox_from_host")

    def debug_placement(self):
        """How device_init placed the grids: {candidates tried, probe ms of the allocation kept, of the slowest one}."""
        n, a, b = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
        self._lib.asora_debug_placement(C.byref(n), C.byref(a), C.byref(b))
        t, pr = C.c_double(0.0), C.c_double(0.0)
        self._lib.asora_debug_init_cost(C.byref(t), C.byref(pr))
        return {"candidates": n.value, "chosen_probe_ms": a.value, "slowest_probe_ms": b.value,
                "device_init_ms": t.value, "probe_cost_ms": pr.value}

    def evolve_slab_fold_all(self):
        _capi.check(self._lib.asora_evolve_slab_fold_all(), "evolve_slab_fold_all")

    def evolve_slab_add(self, i_begin, i_count, dev_ptr):
        """dev_ptr: device address of i_count*N*N doubles (e.g. torch tensor .data_ptr())."""
        _capi.check(self._lib.asora_evolve_slab_add(int(i_begin), int(i_count), C.c_void_p(int(dev_ptr))), "evolve_slab_add")

    def evolve_slab_add_host(self, i_begin, planes):
        a = np.ascontiguousarray(planes, dtype=np.float64)
        _capi.check(self._lib.asora_evolve_slab_add_host(int(i_begin), int(a.shape[0]), _capi.dptr(a)), "evolve_slab_add_host")

    def evolve_slab_pass(self):
        _capi.check(self._lib.asora_evolve_slab_pass(), "evolve_slab_pass")

    def evolve_slab_nhi(self, i_begin, i_count):
        _capi.check(self._lib.asora_evolve_slab_nhi(int(i_begin), int(i_count)), "evolve_slab_nhi")

    def evolve_slab_close(self, sums=None):
        """sums = None: {sum x, sum 1-x, conv_flag} at reduction_ptr() have been summed over the ranks in place; else the
        three sums over all ranks as (conv_flag, sum x, sum 1-x), the order chemistry_finish() returns them in."""
        if sums is None:
            _capi.check(self._lib.asora_evolve_slab_close(None), "evolve_slab_close")
        else:
            a = np.array([float(sums[1]), float(sums[2]), float(sums[0])])
            _capi.check(self._lib.asora_evolve_slab_close(_capi.dptr(a)), "evolve_slab_close")

    def planes_to_host(self, which, i_begin, i_count, N):
        out = np.empty((int(i_count), N, N))
        _capi.check(self._lib.asora_planes_to_host(int(which), int(i_begin), int(i_count), _capi.dptr(out)),
                    "planes_to_host")
        return out

    def planes_to_device(self, which, i_begin, planes):
        a = np.ascontiguousarray(planes, dtype=np.float64)
        _capi.check(self._lib.asora_planes_to_device(int(which), int(i_begin), int(a.shape[0]), _capi.dptr(a)),
                    "planes_to_device")

    def set_option(self, option, value):
        _capi.check(self._lib.asora_set_option(int(option), int(value)), "set_option")

    def get_option(self, option):
        return self._lib.asora_get_option(int(option))

    def kernel_time_ms(self, kernel):
        ms = C.c_double(0.0)
        n = C.c_long(0)
        _capi.check(self._lib.asora_kernel_time_ms(int(kernel), C.byref(ms), C.byref(n)), "kernel_time_ms")
        return ms.value, n.value

    def kernel_time_reset(self):
        self._lib.asora_kernel_time_reset()

    def synchronize(self):
        _capi.check(self._lib.asora_synchronize(), "synchronize")

    def last_raytrace_counts(self):
        g = C.c_longlong(0)
        e = C.c_longlong(0)
        _capi.check(self._lib.asora_last_raytrace_counts(C.byref(g), C.byref(e)), "last_raytrace_counts")
        return g.value, e.value

    def last_raytrace_zero_rates(self):
        """Of the rate-receiving pairs of the last raytrace: how many got exactly +0 and were not added (ASORA_OPT_SKIP_ZERO_RATES)."""
        g, e, z = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _capi.check(self._lib.asora_last_raytrace_counts_ex(C.byref(g), C.byref(e), C.byref(z)), "last_raytrace_counts_ex")
        return z.value

    def last_raytrace_variant(self):
        """{"paired", "aligned", "buffer_atomics", "split_descriptors", "skip_zero", "global_shells": bool, "units", "threads"} of the
        last raytrace launch (asora_last_raytrace_variant)."""
        v = self._lib.asora_last_raytrace_variant()
        return {"paired": bool(v & 1), "aligned": bool(v & 2), "buffer_atomics": bool(v & 4), "split_descriptors": bool(v & 8),
                "skip_zero": bool(v & 16), "global_shells": bool(v & 32), "units": (v >> 8) & 255, "threads": v >> 16}

    def debug_geometry_bytes(self):
        """Device memory of the current geometry tables (shared parts once)."""
        return int(self._lib.asora_debug_geometry_bytes())

    def debug_geometry_tables(self):
        """The geometry tables of the last raytrace launch: (list of (entries x 8) uint32 arrays, dict(nsteps, shells, max_cells, threads))."""
        n, ns, nt, sh, mc, th = C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        _capi.check(self._lib.asora_debug_geometry_table(-1, None, 0, C.byref(n), C.byref(ns), C.byref(nt), C.byref(sh), C.byref(mc),
                                                         C.byref(th)), "debug_geometry_table")
        tables, steps = [], []
        for t in range(nt.value):
            _capi.check(self._lib.asora_debug_geometry_table(t, None, 0, C.byref(n), C.byref(ns), C.byref(nt), C.byref(sh), C.byref(mc),
                                                             C.byref(th)), "debug_geometry_table")
            a = np.zeros((n.value, 8), dtype=np.uint32)
            _capi.check(self._lib.asora_debug_geometry_table(t, a.ctypes.data_as(C.POINTER(C.c_uint32)), n.value, C.byref(n), C.byref(ns),
                                                             C.byref(nt), C.byref(sh), C.byref(mc), C.byref(th)), "debug_geometry_table")
            tables.append(a)
            steps.append(ns.value)
        return tables, {"nsteps": steps, "shells": sh.value, "max_cells": mc.value, "threads": th.value}

    def build_id(self):
        """Hash over the library's sources, headers and compiler flags (asora_build_id)."""
        return self._lib.asora_build_id().decode()

    def debug_coldens(self, R, sig, dr, source_index, N):
        out = np.zeros((N, N, N))
        _capi.check(self._lib.asora_debug_coldens(float(R), float(sig), float(dr), int(source_index),
                                                  _capi.dptr(out), int(N)), "debug_coldens")
        return out


class _Chemistry:
    """Stand-in for ``libc2ray.chemistry`` (f2py wrapper of src/c2ray/chemistry.f90)."""

    def __init__(self, lib):
        self._lib = lib

    def global_pass(self, dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c):
        """conv_flag = global_pass(...); xh_av and xh_intermed are updated IN PLACE, xh is not
        modified (chemistry.f90:13-48,107-108).  The pass is elementwise, so all grids are brought to
        the storage order of xh_av; like f2py's intent(inout), xh_av and xh_intermed must be
        float64 arrays."""
        for name, a in (("xh_av", xh_av), ("xh_intermed", xh_intermed)):
            if not isinstance(a, np.ndarray) or a.dtype != np.float64:
                raise ValueError(f"{name} must be a float64 array (updated in place)")
        shape = np.shape(xh_av)
        if len(shape) != 3:
            raise ValueError("grids must be 3-dimensional")
        order = 'F' if (xh_av.flags.f_contiguous and not xh_av.flags.c_contiguous) else 'C'
        req = lambda a: np.require(np.broadcast_to(np.asarray(a, dtype=np.float64), shape), np.float64, [order])
        nd, tp, x0, ph = req(ndens), req(temp), req(xh), req(phi_ion)
        xa = np.require(xh_av, np.float64, [order, 'W'])
        xi = np.require(xh_intermed, np.float64, [order, 'W'])
        if xi is xh_intermed and np.shares_memory(xa, xi):
            xi = xi.copy(order=order)          # aliased in/out grids (hydrogenODE): keep them apart
        if np.shares_memory(x0, xa) or np.shares_memory(x0, xi):
            x0 = x0.copy(order=order)
        conv = C.c_int(0)
        _capi.check(self._lib.c2ray_global_pass(float(dt), _capi.dptr(nd), _capi.dptr(tp), _capi.dptr(x0),
                                                _capi.dptr(xa), _capi.dptr(xi), _capi.dptr(ph), float(bh00),
                                                float(albpow), float(colh0), float(temph0), float(abu_c),
                                                int(shape[0]), int(shape[1]), int(shape[2]), C.byref(conv)),
                    "global_pass")
        if xa is not xh_av:
            xh_av[...] = xa
        if xi is not xh_intermed:
            xh_intermed[...] = xi
        return conv.value


class _Raytracing:
    """Stand-in for ``libc2ray.raytracing`` (f2py wrapper of src/c2ray/raytracing.f90)."""

    def __init__(self, lib):
        self._lib = lib

    def do_all_sources(self, normflux, srcpos, max_subbox, subboxsize, coldensh_out, sig, dr, ndens, xh_av,
                       phi_ion, phi_heat, loss_fraction, photo_thin_table, photo_thick_table,
                       heat_thin_table, heat_thick_table, minlogtau, dlogtau, r_max_lls):
        """sum_nbox, photon_loss = do_all_sources(...)   (f2py signature of raytracing.f90:52-56)

        The reference's CPU raytracer -- cubic sub-boxes grown until the photon loss through the box faces is
        small, rates within r_max_lls, heating rates, column densities of the last source -- evaluated on the
        GPU (csrc/subbox.hip).  As with f2py's intent(inout), coldensh_out, phi_ion and phi_heat must be
        Fortran-contiguous float64 (N,N,N) arrays and are updated in place; ndens and xh_av are copied to
        Fortran order when needed; srcpos is (3,NumSrc), 1-based."""
        flux = np.ascontiguousarray(normflux, dtype=np.float64)
        pos = np.asfortranarray(srcpos, dtype=np.int32)
        if pos.ndim != 2 or pos.shape[0] != 3 or pos.shape[1] != flux.size:
            raise ValueError("srcpos must have shape (3, NumSrc) matching normflux")
        shape = np.shape(coldensh_out)
        if len(shape) != 3 or shape[0] != shape[1] or shape[0] != shape[2]:
            raise ValueError("grids must have shape (N,N,N)")
        for name, a in (("coldensh_out", coldensh_out), ("phi_ion", phi_ion), ("phi_heat", phi_heat)):
            if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.shape != shape or not a.flags.f_contiguous:
                raise ValueError(f"{name} must be a Fortran-contiguous float64 array of shape {shape} "
                                 "(updated in place)")
        nd = np.asfortranarray(ndens, dtype=np.float64)
        xa = np.asfortranarray(xh_av, dtype=np.float64)
        if nd.shape != shape or xa.shape != shape:
            raise ValueError("ndens and xh_av must have the shape of the output grids")
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in
                (photo_thin_table, photo_thick_table, heat_thin_table, heat_thick_table)]
        numtau = tabs[0].size
        if any(t.size != numtau for t in tabs):
            raise ValueError("the four radiation tables must have the same length")
        nbox = C.c_int(0)
        loss = C.c_double(0.0)
        N = int(shape[0])
        _capi.check(self._lib.c2ray_do_all_sources(
            _capi.dptr(flux), _capi.iptr(pos), int(max_subbox), int(subboxsize), _capi.dptr(coldensh_out),
            float(sig), float(dr), _capi.dptr(nd), _capi.dptr(xa), _capi.dptr(phi_ion), _capi.dptr(phi_heat),
            float(loss_fraction), _capi.dptr(tabs[0]), _capi.dptr(tabs[1]), _capi.dptr(tabs[2]), _capi.dptr(tabs[3]),
            float(minlogtau), float(dlogtau), float(r_max_lls), int(numtau), int(flux.size), N, N, N,
            C.byref(nbox), C.byref(loss)), "do_all_sources")
        return nbox.value, loss.value


class _LibC2Ray:
    """Stand-in for the reference's f2py module ``libc2ray`` (sub-modules chemistry, raytracing)."""

    def __init__(self, lib):
        self.chemistry = _Chemistry(lib)
        self.raytracing = _Raytracing(lib)


_c2ray_lib = None
_asora_lib = None


def load_c2ray():
    """pyc2ray/load_extensions.py:9-29.  Raises RuntimeError when the library is missing."""
    global _c2ray_lib
    if _c2ray_lib is None:
        _c2ray_lib = _LibC2Ray(_capi.load())
    return _c2ray_lib


def load_asora():
    """pyc2ray/load_extensions.py:31-48.  Unlike the reference, a missing library is an error."""
    global _asora_lib
    if _asora_lib is None:
        _asora_lib = _LibAsora(_capi.load())
    return _asora_lib
