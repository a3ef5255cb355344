"""TEST INFRASTRUCTURE: a host-memory stand-in for the ctypes `libasora` object, backed by the CPU
oracle.  It lets the world_size-2 gloo tests exercise the product's source sharding, all-reduce and
convergence logic (pyc2ray_amd/evolve.py, pyc2ray_amd/dist.py) on a machine without a GPU.  It is
injected by monkeypatching inside tests only; the product never imports it."""
import numpy as np

from oracle import oracle as O


class OracleAsora:
    def __init__(self, thin, thick):
        self.thin, self.thick = thin, thick
        self.g = {}
        self.pos = self.flux = None

    def source_data_to_device(self, pos, flux, n):
        self.pos, self.flux = np.array(pos[:3 * n]), np.array(flux[:n])

    def grid_to_device(self, which, a):
        self.g[which] = np.ascontiguousarray(a, dtype=np.float64).copy()

    def grid_copy(self, dst, src):
        self.g[dst] = self.g[src].copy()

    def grid_to_host(self, which, out):
        out[...] = self.g[which]
        return out

    def grid_sum(self, which):
        return float(self.g[which].sum())

    def host_empty(self, shape, order='C'):
        return np.empty(shape, order=order)

    def device_ptr(self, which):
        return 0

    def synchronize(self):
        pass

    def raytrace_device(self, R, sig, dr, src_begin, src_count, minlogtau, dlogtau, NumTau):
        N = self.g[0].shape[0]
        if src_count == 0:
            self.g[2] = np.zeros((N, N, N))
            return
        sl = slice(3 * src_begin, 3 * (src_begin + src_count))
        self.g[2] = O.asora_do_all_sources(R, sig, dr, self.g[0], self.g[1], self.pos[sl],
                                           self.flux[src_begin:src_begin + src_count], self.thin, self.thick,
                                           minlogtau, dlogtau, NumTau=NumTau, flags=O.ASORA_MODE)["phi_ion"]

    # the sub-box raytracer on resident grids (evolve3D with use_gpu=False)
    def device_init_auto(self, N):
        pass

    def photo_table_to_device(self, thin, thick, NumTau):
        self.thin, self.thick = np.array(thin), np.array(thick)

    def subbox_raytrace_device(self, max_subbox, subboxsize, loss_fraction, R, sig, dr, minlogtau, dlogtau, NumTau,
                               src_begin, src_count):
        N = self.g[0].shape[0]
        if src_count == 0:
            self.g[2] = np.zeros((N, N, N))
            return 0, 0.0
        pos1 = self.pos[3 * src_begin:3 * (src_begin + src_count)].reshape(src_count, 3).T + 1
        r = O.do_all_sources(self.flux[src_begin:src_begin + src_count], pos1, max_subbox, subboxsize, sig, dr, self.g[0],
                             self.g[1], loss_fraction, self.thin, self.thick, minlogtau, dlogtau, R, NumTau=NumTau)
        self.g[2] = np.ascontiguousarray(r["phi_ion"])
        return r["nsubbox"], r["photon_loss"]

    # the three-part raytrace of the pipelined path
    def raytrace_begin(self, R, sig, dr, minlogtau, dlogtau, NumTau):
        self._planes_mode = False
        N = self.g[0].shape[0]
        self._rt = (R, sig, dr, minlogtau, dlogtau, NumTau)
        self.g[2] = np.zeros((N, N, N))
        self._folded = np.zeros(N, dtype=bool)

    def raytrace_begin_planes(self, R, sig, dr, minlogtau, dlogtau, NumTau, runs):
        """Like the library: the accumulator is zeroed and nHI is formed on the given planes ONLY.  nHI of every other
        plane is whatever an earlier call left there -- NaN if none did, so a plan that forgets planes a source
        reaches poisons the result instead of passing by accident."""
        N = self.g[0].shape[0]
        self._rt = (R, sig, dr, minlogtau, dlogtau, NumTau)
        if 2 not in self.g or self.g[2].shape != (N, N, N):
            self.g[2] = np.full((N, N, N), np.nan)
        if getattr(self, "_nhi", None) is None or self._nhi.shape != (N, N, N):
            self._nhi = np.full((N, N, N), np.nan)
        for a, cnt in runs:
            self.g[2][a:a + cnt] = 0.0
            self._nhi[a:a + cnt] = self.g[0][a:a + cnt] * (1.0 - self.g[1][a:a + cnt])
        self._folded = np.zeros(N, dtype=bool)
        self._planes_mode = True

    def planes_to_host(self, which, i_begin, i_count, N):
        return self.g[which][i_begin:i_begin + i_count].copy()

    def planes_to_device(self, which, i_begin, planes):
        planes = np.asarray(planes)
        self.g[which][i_begin:i_begin + planes.shape[0]] = planes.reshape(planes.shape[0], *self.g[which].shape[1:])

    def raytrace_range(self, src_begin, src_count):
        if src_count == 0:
            return
        R, sig, dr, minlogtau, dlogtau, NumTau = self._rt
        sl = slice(3 * src_begin, 3 * (src_begin + src_count))
        if getattr(self, "_planes_mode", False):
            # nHI as formed by raytrace_begin_planes (density = nHI, ionised fraction = 0)
            add = O.asora_do_all_sources(R, sig, dr, self._nhi, np.zeros_like(self._nhi), self.pos[sl],
                                         self.flux[src_begin:src_begin + src_count], self.thin, self.thick,
                                         minlogtau, dlogtau, NumTau=NumTau, flags=O.ASORA_MODE)["phi_ion"]
            w = add != 0
            self.g[2][w] += add[w]
            return
        self.g[2] = self.g[2] + O.asora_do_all_sources(R, sig, dr, self.g[0], self.g[1], self.pos[sl],
                                                       self.flux[src_begin:src_begin + src_count], self.thin,
                                                       self.thick, minlogtau, dlogtau, NumTau=NumTau,
                                                       flags=O.ASORA_MODE)["phi_ion"]

    def raytrace_fold(self, i_begin, i_count):
        assert not self._folded[i_begin:i_begin + i_count].any(), "a plane was folded twice"
        self._folded[i_begin:i_begin + i_count] = True

    def stream_ptr(self):
        return 0

    def chemistry_range(self, dt, bh00, albpow, colh0, temph0, abu_c, i_begin, i_count, first):
        if first:
            self._red = [0, 0.0, 0.0]
        sl = slice(i_begin, i_begin + i_count)
        if i_count == 0:
            return
        xa, xi, conv, _ = O.global_pass(dt, self.g[0][sl], self.g[3][sl], self.g[4][sl], self.g[1][sl], self.g[5][sl],
                                        self.g[2][sl], bh00, albpow, colh0, temph0, abu_c)
        self.g[1][sl], self.g[5][sl] = xa, xi
        self._red[0] += conv
        self._red[1] += float(np.sum(xi))
        self._red[2] += float(np.sum(1.0 - xi))

    def chemistry_finish(self):
        return tuple(self._red)

    # ---- the device-resident loop, sharded (asora_evolve_slab_*): same call sequence, numpy underneath.  Wherever the library
    # would hold stale data, the stand-in holds NaN, so a plan that forgets a plane poisons the result instead of passing by
    # accident: after every iteration nHI and xh_av of every plane this rank does not own are NaN until the exchange has
    # delivered xh_av there and evolve_slab_nhi has formed nHI from it.
    def evolve_begin_slab(self, dt, bh00, albpow, colh0, temph0, abu_c, R, sig, dr, minlogtau, dlogtau, NumTau,
                          src_begin, src_count, conv_criterion, convergence_fraction, own_begin, own_count):
        N = self.g[0].shape[0]
        self._ev = dict(chem=(dt, bh00, albpow, colh0, temph0, abu_c), rt=(R, sig, dr, minlogtau, dlogtau, NumTau),
                        src=(src_begin, src_count), crit=conv_criterion, frac=convergence_fraction,
                        own=slice(own_begin, own_begin + own_count), niter=0, done=False, reported=0, rows=[],
                        prev1=2.0 * N ** 3, prev0=2.0 * N ** 3, passed=False)
        self._nhi = self.g[0] * (1.0 - self.g[4])            # from xh: xh_av = copy(xh), evolve.py:136
        self.g[1] = np.full((N, N, N), np.nan)               # XH_AV: nothing valid yet
        self.g[5] = self.g[4].copy()
        self._acc = np.zeros((N, N, N))
        self._outbox = np.full((N, N, N), np.nan)
        self._phi_own = np.zeros((N, N, N))
        self._folded = np.zeros(N, dtype=bool)
        self._first = True
        self._rates_in_outbox = False

    def evolve_slab_fold_all(self):
        """All planes into the out-box (the full-grid exchange on the same loop); the pass then reads the out-box."""
        e = self._ev
        assert not e["passed"] and e["own"].start == 0 and e["own"].stop == self.g[0].shape[0]
        self._rates_in_outbox = True
        if e["done"]:
            return
        self._outbox = self._acc.copy()
        self._acc[:] = 0.0

    def evolve_slab_outbox_from_host(self, i_begin, planes):
        planes = np.asarray(planes)
        self._outbox[i_begin:i_begin + planes.shape[0]] = planes.reshape(planes.shape[0], *self._outbox.shape[1:])

    def evolve_slab_trace(self, src_begin, src_count):
        e = self._ev
        if e["done"] or src_count == 0:
            return
        assert not e["passed"]
        R, sig, dr, minlogtau, dlogtau, NumTau = e["rt"]
        sl = slice(3 * src_begin, 3 * (src_begin + src_count))
        add = O.asora_do_all_sources(R, sig, dr, self._nhi, np.zeros_like(self._nhi), self.pos[sl],
                                     self.flux[src_begin:src_begin + src_count], self.thin, self.thick,
                                     minlogtau, dlogtau, NumTau=NumTau, flags=O.ASORA_MODE)["phi_ion"]
        w = add != 0
        self._acc[w] += add[w]

    def evolve_slab_fold_out(self, i_begin, i_count):
        if self._ev["done"]:
            return
        sl = slice(i_begin, i_begin + i_count)
        assert not self._folded[sl].any(), "a plane was folded twice"
        own = self._ev["own"]
        assert i_begin + i_count <= own.start or i_begin >= own.stop or i_count == 0, "an own plane was sent away"
        self._folded[sl] = True
        self._outbox[sl] = self._acc[sl]
        self._acc[sl] = 0.0

    def evolve_slab_outbox_to_host(self, i_begin, i_count, N):
        return self._outbox[i_begin:i_begin + i_count].copy()

    def evolve_slab_add_host(self, i_begin, planes):
        if self._ev["done"]:
            return
        planes = np.asarray(planes)
        own = self._ev["own"]
        assert own.start <= i_begin and i_begin + planes.shape[0] <= own.stop, "rates received for a plane this rank does not own"
        self._acc[i_begin:i_begin + planes.shape[0]] += planes.reshape(planes.shape[0], *self._acc.shape[1:])

    def evolve_slab_pass(self):
        e = self._ev
        e["passed"] = True
        self._red = [0, 0.0, 0.0]
        if e["done"]:
            return
        sl = e["own"]
        if sl.stop > sl.start:
            xav_in = self.g[4][sl] if self._first else self.g[1][sl]
            rates = self._outbox if self._rates_in_outbox else self._acc
            xa, xi, conv, _ = O.global_pass(e["chem"][0], self.g[0][sl], self.g[3][sl], self.g[4][sl], xav_in, self.g[5][sl],
                                            rates[sl], *e["chem"][1:])
            self.g[1][sl], self.g[5][sl] = xa, xi
            self._phi_own[sl] = rates[sl]
            self._acc[sl] = 0.0
            self._nhi[sl] = self.g[0][sl] * (1.0 - xa)
            self._red = [conv, float(np.sum(xi)), float(np.sum(1.0 - xi))]
        # what the next trace may read must be formed from data that ARRIVES after this pass: poison everything not owned
        keep = np.zeros(self._nhi.shape[0], dtype=bool)
        keep[sl] = True
        self._nhi[~keep] = np.nan
        self.g[1][~keep] = np.nan

    def evolve_slab_nhi(self, i_begin, i_count):
        if self._ev["done"]:
            return
        sl = slice(i_begin, i_begin + i_count)
        self._nhi[sl] = self.g[0][sl] * (1.0 - self.g[1][sl])

    def evolve_slab_close(self, sums=None):
        e = self._ev
        assert e["passed"] and (sums is not None or self._rates_in_outbox)
        if sums is None:
            sums = self._red                 # (every rank passed over the whole grid: the sums are the totals already)
        e["passed"] = False
        self._folded[:] = False
        if e["done"]:
            return
        nconv, s1, s0 = float(sums[0]), float(sums[1]), float(sums[2])
        rel1 = abs((s1 - e["prev1"]) / s1) if s1 > 0.0 else 1.0
        rel0 = abs((s0 - e["prev0"]) / s0) if s0 > 0.0 else 1.0
        e["rows"].append((nconv, s1, s0, rel1, rel0))
        e["prev1"], e["prev0"] = s1, s0
        e["niter"] += 1
        e["done"] = (nconv < e["crit"]) or (rel1 < e["frac"] and rel0 < e["frac"])
        self._first = False

    def evolve_poll(self, max_rows=32):
        e = self._ev
        rows = e["rows"][e["reported"]:e["reported"] + max_rows] if max_rows > 0 else []
        e["reported"] = e["reported"] + len(rows) if max_rows > 0 else e["niter"]
        N = self.g[0].shape[0]
        phi = np.full((N, N, N), np.nan)
        phi[e["own"]] = self._phi_own[e["own"]]
        self.g[2] = phi
        return e["niter"], e["done"], np.array(rows).reshape(-1, 5)

    def chemistry_device(self, dt, bh00, albpow, colh0, temph0, abu_c):
        xa, xi, conv, _ = O.global_pass(dt, self.g[0], self.g[3], self.g[4], self.g[1], self.g[5], self.g[2],
                                        bh00, albpow, colh0, temph0, abu_c)
        self.g[1], self.g[5] = xa, xi
        return conv, float(np.sum(xi)), float(np.sum(1.0 - xi))


class OracleC2Ray:
    """Stand-in for the `libc2ray` object (raytracing.do_all_sources, chemistry.global_pass), backed by the oracle."""

    class _Raytracing:
        def do_all_sources(self, normflux, srcpos, max_subbox, subboxsize, coldensh_out, sig, dr, ndens, xh_av, phi_ion,
                           phi_heat, loss_fraction, thin, thick, heat_thin, heat_thick, minlogtau, dlogtau, r_max_lls):
            r = O.do_all_sources(normflux, srcpos, max_subbox, subboxsize, sig, dr, ndens, xh_av, loss_fraction, thin, thick,
                                 minlogtau, dlogtau, r_max_lls, heat_thin=heat_thin, heat_thick=heat_thick)
            coldensh_out[...] = r["coldens"]
            phi_ion[...] = r["phi_ion"]
            phi_heat[...] += r["phi_heat"]
            return r["nsubbox"], r["photon_loss"]

    class _Chemistry:
        def global_pass(self, dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c):
            xa, xi, conv, _ = O.global_pass(dt, ndens, temp, xh, xh_av, xh_intermed, phi_ion, bh00, albpow, colh0, temph0, abu_c)
            xh_av[...] = xa
            xh_intermed[...] = xi
            return conv

    def __init__(self):
        self.raytracing = self._Raytracing()
        self.chemistry = self._Chemistry()
