#!/usr/bin/env python3
"""bench.py -- throughput of the pyc2ray hot path (ASORA raytrace + photo-ionisation chemistry) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N --steps K --warmup W        (N > 1, no launcher: bench.py starts the line above as a CHILD process
                                                        before it touches the GPU, relays the JSON line and the exit code)

One "step" = one outer iteration of evolve3D (pyc2ray/evolve.py:168-240) on device-resident inputs, in its steady
state: raytrace all of this rank's sources, [N>1: exchange the rates between ranks,] one chemistry pass with its
convergence reductions -- the same launches evolve3D makes per iteration (asora_evolve_enqueue on one GPU,
TorchComm.slab_enqueue on several).  The convergence test is evaluated but can never pass (criterion -1), so
every step does its full work.  The FIRST iteration of a time step additionally forms nHI from xh and zeroes the
accumulators; its time is reported beside (config.first_iteration_of_a_time_step_ms).

Workload at N=1 (BASELINE.json configs[2], the configuration the metric is quoted on): 256^3 grid, uniform
ndens = 1e-3, xh = 2e-4, T = 1e4 K, dr = 3*3.086e24/256 cm, 1000 equal sources at RandomState(100) positions,
R = 32 cells, black-body Teff = 1e5 K table with NumTau = 20000.
At N>1 the default is the metric's multi-GPU mode, BASELINE configs[3]: the same 1000 sources IN TOTAL, on the
densest cells of a log-normal 256^3 density, sharded over the ranks in contiguous blocks (pyc2ray/evolve.py:362-367)
of the list ordered by first coordinate -- STRONG scaling ("scaling": "strong").  The rates are exchanged plane-wise
(pyc2ray_amd/dist.py SlabPlan: rates to the owners of the planes, slab chemistry, xh_av back); `--exchange allreduce`
selects the full-grid RCCL all-reduce with replicated chemistry instead.  `--scaling weak` gives every rank --nsrc
sources (the global list then has gpus*nsrc); `--workload` overrides the density/source model.

Unit of work ("cell-update"):
  raytrace  = one (source, cell) pair that receives a rate: |d| <= R inside the periodic window
              (the set src/asora/raytracing.cu:315 admits).  NOT the larger set of scratch
              evaluations the reference makes (clipped octahedron, 1.75x more) and not this build's
              own evaluation count (which includes octant-boundary planes twice).
  chemistry = one cell in one global_pass (N^3 per step).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # multi-process GPU work on this pool needs dmabuf IPC

MYR = 3.15576e13
SIG = 6.30e-18
MINLOGTAU, MAXLOGTAU, NUMTAU = -20.0, 4.0, 20000
BH00, ALBPOW, ABU_C = 2.59e-13, -0.7, 7.1e-7
COLH0 = 1.3e-8 * 0.83 * 1.0 / 13.598 ** 2          # colh0_fact*fh0*xih0/eth0^2 (c2ray_base.py:346)
TEMPH0 = 13.598 / 8.617e-05                        # eth0*ev2k                   (c2ray_base.py:77,347)
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
RT_BYTES_PER_UPDATE = 32                           # 8 ndens + 8 xh_av + 16 Gamma RMW
CHEM_BYTES_PER_UPDATE = 56                         # 5 loads + 2 stores
CHEM_FUSED_BYTES_PER_UPDATE = 88                   # the fused pass on a uniform-temperature grid (both workloads here): 5 loads
                                                   # + 6 stores; 96 with a temperature grid that has to be read


def make_tables(numtau=NUMTAU):
    """Teff = 1e5 K black-body tables over log10(tau) in [-20, 4].  numtau = 20000: the benchmark's parameter file
    (ref: test/paper_tests/raytracing_benchmark/parameters.yml:73-77); 2000: the reference's production parameter files
    (ref: test/paper_eor_simulation/parameters.yml:75)."""
    from pyc2ray_amd.radiation import BlackBodySource, make_tau_table
    ev2fr = 0.241838e15
    tau, dlog = make_tau_table(MINLOGTAU, MAXLOGTAU, numtau)
    src = BlackBodySource(1e5, False, ev2fr * 13.598, 2.8)
    thin, thick = src.make_photo_table(tau, ev2fr * 13.598, 10 * ev2fr * 54.416, 1e48)
    return thin, thick, dlog


def make_workload(kind, N, nsrc_total):
    """Returns ndens, xh, temp, dr, src_pos (3,ns) 1-based, src_flux."""
    temp = np.full((N, N, N), 1e4)
    xh = np.full((N, N, N), 2e-4)
    if kind == "uniform":
        ndens = np.full((N, N, N), 1e-3)
        dr = 3 * 3.086e24 / N
        rng = np.random.RandomState(100)
        pos = (1 + rng.randint(0, N, size=3 * nsrc_total)).reshape((nsrc_total, 3), order="C").T.copy()
        flux = np.ones(nsrc_total)
    elif kind == "cosmo":
        rng = np.random.default_rng(2024)
        white = rng.normal(size=(N, N, N))
        k = np.fft.fftfreq(N)
        k2 = k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :N // 2 + 1] ** 2
        k2[0, 0, 0] = 1.0
        g = np.fft.irfftn(np.fft.rfftn(white) / k2 ** 0.5 * (k2 > 0), s=(N, N, N), axes=(0, 1, 2))   # P(k) ~ k^-2
        g /= g.std()
        sigma, nbar = 1.2, 1.87e-7 * (1 + 9.938) ** 3
        ndens = nbar * np.exp(sigma * g - sigma ** 2 / 2)
        dr = 3 * 3.086e24 / N / (1 + 9.938)
        idx = np.argsort(ndens, axis=None)[::-1][:nsrc_total]
        pos = np.array(np.unravel_index(idx, (N, N, N))) + 1
        flux = ndens.ravel()[idx]
        flux = flux / flux.mean()
    else:
        raise ValueError(kind)
    return ndens, xh, temp, dr, pos, flux


def workload_label(kind, N, nsrc_total, R, world, strong):
    """Names the BASELINE.json config a run corresponds to (configs[2] is what `metric` is quoted on at one GPU,
    configs[3] its multi-GPU mode)."""
    share = (f"{nsrc_total} sources in total, sharded over {world} GPUs" if (world > 1 and strong) else
             f"{nsrc_total // max(world, 1)} sources per GPU x {world} GPUs" if world > 1 else f"{nsrc_total} sources")
    if kind == "uniform":
        tag = "BASELINE configs[2]: " if (N, nsrc_total, world) == (256, 1000, 1) and R in (16.0, 32.0, 64.0) else ""
        return (f"{tag}{N}^3 uniform ndens=1e-3 xh=2e-4, {share} at random positions, r_RT={R:g}, "
                "raytrace + one chemistry pass per step")
    tag = ("BASELINE configs[3]: " if (N, nsrc_total, R) == (256, 1000, 32.0) and (world == 1 or strong) else
           "BASELINE configs[4] on one GPU: " if (N, nsrc_total, R, world) == (512, 100000, 32.0, 1) else "")
    return (f"{tag}{N}^3 log-normal density, {share} on the densest cells, r_RT={R:g}, "
            "raytrace + one chemistry pass per step")


# 64-B atomic requests per second the memory side takes from no-return global_atomic_add_f64 (rows of 8 ... 4096 doubles at any
# alignment, scattered over 2 x 256^3 doubles): tools/micro/atomic_rate.hip, measured on MI355X
ATOMIC_REQUEST_CEILING = 2.28e10
ATOMIC_CEILING_SOURCE = os.path.join("profiles", "r04_atomic_rate_microbench.txt")


def find_pmc_summary(build_id):
    """(path relative to the repository, None) of the newest committed counter summary collected ON THIS BUILD of the library
    (profiles/rNN_pmc_summary*.txt whose header holds `# build_id <asora_build_id()>`; tools/pmc.sh writes it), or
    (None, reason).  A summary of another build is never paired with this run's timings: editing a kernel without re-running
    tools/pmc.sh makes `traffic` null, with the reason in the JSON."""
    import glob
    import re
    cands = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_summary*.txt")):
        ident = None
        with open(path) as f:
            for line in f:
                if not line.startswith("#"):
                    break
                m = re.match(r"#\s*build_id\s+(\S+)", line)
                if m:
                    ident = m.group(1)
        rnd = int(re.match(r"r(\d+)", os.path.basename(path)).group(1))
        cands.append((rnd, os.path.getmtime(path), path, ident))
    match = sorted(c for c in cands if c[3] == build_id)
    if match:
        return os.path.relpath(match[-1][2], ROOT), None
    if not cands:
        return None, "no counter summary under profiles/"
    newest = sorted(cands)[-1]
    return None, (f"no committed counter summary was collected on this build of the library (asora_build_id {build_id}); the newest, "
                  f"{os.path.relpath(newest[2], ROOT)}, names build {newest[3] or 'none (collected before build ids existed)'}: "
                  "re-run tools/pmc.sh on the current library")


def pmc_counters(kernel, summary):
    """Mean per launch of every counter the counter summary `summary` holds for `kernel` (rocprofv3 --pmc passes of this same
    command, one counter group per pass; tools/pmc.sh)."""
    out = {}
    if not summary:
        return out
    for line in open(os.path.join(ROOT, summary)):
        parts = line.split()
        if len(parts) >= 4 and parts[0].startswith(kernel):
            out[parts[1]] = float(parts[-1].split("=")[1])
    return out


def pmc_traffic_bytes(kernel, summary):
    """HBM bytes per launch of `kernel`: FETCH_SIZE and WRITE_SIZE collected in separate passes; the counters are in
    KiB and FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 -- the doubling was
    checked on this repository's streaming chemistry kernel, whose 2*FETCH_SIZE equals its N^3 float64 loads exactly.
    None when the summary does not hold the counters."""
    c = pmc_counters(kernel, summary)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0


def cpu_baseline(kind, N, ndens, xh, temp, dr, pos, flux, thin, thick, dlog, R, nsrc_job, budget_sources):
    """The reference's own CPU path (flang-built src/c2ray/*.f90, oracle/_ref) -- or the oracle's C
    restatement when that build is absent -- on a bounded sample of the same workload:
    the first `budget_sources` sources (cube +-R around each, as the Fortran sweeps it) and one
    global_pass over the full grid; extrapolated linearly in the number of sources to the job."""
    from oracle import ref_fortran as F
    from oracle import oracle as O
    use_ref = F.available()
    mod = F if use_ref else O
    ns = min(budget_sources, pos.shape[1])
    Ri = int(np.ceil(R))
    nd_f = np.asfortranarray(ndens)
    xh_f = np.asfortranarray(xh)
    t0 = time.time()
    r = mod.do_all_sources(flux[:ns], pos[:, :ns], max_subbox=Ri, subboxsize=Ri, sig=SIG, dr=dr, ndens=nd_f,
                           xh_av=xh_f, loss_fraction=0.0, thin=thin, thick=thick, minlogtau=MINLOGTAU,
                           dlogtau=dlog, R_max_LLS=R, NumTau=thin.shape[0] - 1)
    t_rt = time.time() - t0
    t0 = time.time()
    mod.global_pass(MYR, nd_f, np.asfortranarray(temp), xh_f, xh_f, xh_f, r["phi_ion"], BH00, ALBPOW, COLH0,
                    TEMPH0, ABU_C)
    t_chem = time.time() - t0
    return use_ref, ns, t_rt, t_chem


# ---- one socket's cores: independent single-threaded processes over sources (the reference is single-threaded,
# raytracing.f90:177, and sources are independent)
def _cpu_worker():
    """`python bench.py --cpu-worker`: one host core's share of the CPU sample.  Reads its job (one JSON line), maps the
    inputs -- ndens, xh, temp as READ-ONLY memory maps of the files the parent wrote once (shared by all workers: a worker
    owns only the three N^3 output grids the reference's routine fills) --, reports "ready", waits for "go", then times the
    reference raytracer on its sources and one global_pass on its slab of the grid.  Never touches the GPU."""
    job = json.loads(sys.stdin.readline())
    from oracle import ref_fortran as F
    from oracle import oracle as O
    use_ref = F.available()
    mod = F if use_ref else O
    d = job["dir"]
    ndens, xh, temp = (np.load(os.path.join(d, n + ".npy"), mmap_mode="r") for n in ("ndens", "xh", "temp"))
    meta = np.load(os.path.join(d, "meta.npz"))
    thin, thick, dlog, dr, pos, flux = meta["thin"], meta["thick"], float(meta["dlog"]), float(meta["dr"]), meta["pos"], meta["flux"]
    lo, hi = job["sources"]
    a, b = job["planes"]
    sl = lambda g: np.asfortranarray(g[a:b])
    Ri = int(np.ceil(job["R"]))
    extra = {"copy_xh_av": False} if use_ref else {}
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.time()
    r = mod.do_all_sources(flux[lo:hi], pos[:, lo:hi], max_subbox=Ri, subboxsize=Ri, sig=SIG, dr=dr, ndens=ndens,
                           xh_av=xh, loss_fraction=0.0, thin=thin, thick=thick, minlogtau=MINLOGTAU, dlogtau=dlog,
                           R_max_LLS=job["R"], NumTau=thin.shape[0] - 1, **extra)
    t1 = time.time()
    if b > a:
        mod.global_pass(MYR, sl(ndens), sl(temp), sl(xh), sl(xh), sl(xh), sl(r["phi_ion"]), BH00, ALBPOW, COLH0, TEMPH0, ABU_C)
    t2 = time.time()
    print(json.dumps({"t_rt": t1 - t0, "t_chem": t2 - t1, "sources": hi - lo, "reference": bool(use_ref)}), flush=True)


def usable_memory_bytes():
    """What this process may still allocate: MemAvailable, or less when the control group says so."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read())
        if lim != "max":
            room = int(lim) - cur
            avail = room if avail is None else min(avail, room)
    except (OSError, ValueError):
        pass
    return avail


def start_cpu_workers(cores, workload, N, R, sample_sources, tables):
    """Started BEFORE anything initialises the GPU in this process (no exec afterwards).  `workload` = (ndens, xh, temp, dr, pos,
    flux): written once, Fortran-ordered, to a directory in shared memory that every worker maps read-only."""
    import subprocess
    import tempfile
    ndens, xh, temp, dr, pos, flux = workload
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="pyc2ray_amd_bench_", dir=shm)
    for name, g in (("ndens", ndens), ("xh", xh), ("temp", temp)):
        np.save(os.path.join(d, name + ".npy"), np.asfortranarray(g))
    np.savez(os.path.join(d, "meta.npz"), thin=tables[0], thick=tables[1], dlog=tables[2], dr=dr, pos=pos, flux=flux)
    procs = []
    # whatever happens to this process afterwards: no worker and no file in shared memory (it is held in RAM) is left behind
    import atexit
    import shutil

    def _cleanup():
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(d, ignore_errors=True)
    atexit.register(_cleanup)
    for w in range(cores):
        job = {"dir": d, "N": N, "R": R,
               "sources": [w * sample_sources // cores, (w + 1) * sample_sources // cores],
               "planes": [w * N // cores, (w + 1) * N // cores]}
        pr = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker"], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, text=True, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"))
        pr.stdin.write(json.dumps(job) + "\n")
        pr.stdin.flush()
        procs.append(pr)
    return procs, d


def run_cpu_workers(procs):
    """Wall-clock time of the sample on all workers at once; returns (wall_rt, wall_total, results)."""
    for pr in procs:
        if pr.stdout.readline().strip() != "ready":
            raise RuntimeError("cpu worker failed to start")
    t0 = time.time()
    for pr in procs:
        pr.stdin.write("go\n")
        pr.stdin.flush()
    res = [json.loads(pr.stdout.readline()) for pr in procs]
    wall = time.time() - t0
    for pr in procs:
        pr.stdin.close()
        pr.wait(timeout=60)
    return wall, res


def trip_counts(dt, n, T, x0, xav, gamma, bh00, albpow, colh0, temph0, abu_c):
    """do_chemistry's trip count per cell (chemistry.f90:146-203 with doric :279-311), vectorised on the host: how many times
    each cell goes round the loop the fused pass runs per lane (a wave lasts as long as its slowest lane)."""
    brech0 = bh00 * (T / 1e4) ** albpow
    acolh0 = colh0 * np.sqrt(T) * np.exp(-temph0 / T)
    nit = np.zeros(n.shape, dtype=np.int32)
    live = np.ones(n.shape, dtype=bool)
    xav = xav.copy()
    for it in range(1, 402):
        idx = np.flatnonzero(live)
        if idx.size == 0:
            break
        nn, xa, g = n.flat[idx], xav.flat[idx], gamma.flat[idx]
        de = nn * (xa + abu_c)
        aih0 = g + de * acolh0.flat[idx]
        delth = aih0 + de * brech0.flat[idx]
        eqxh = aih0 / delth
        deltht = delth * dt
        ee = np.exp(-deltht)
        avg = np.where(deltht < 1.0e-8, 1.0, (1.0 - ee) / np.where(deltht == 0, 1.0, deltht))
        new = np.maximum(eqxh + (x0.flat[idx] - eqxh) * avg, 1e-14)
        nit.flat[idx] = it
        done = (np.abs((new - xa) / (1.0 - new)) < 1.0e-3) | (1.0 - new < 1.0e-8) | (it > 400)
        xav.flat[idx] = new
        live.flat[idx[done]] = False
    return nit


def evolving_state(lib, p, _capi, N, R, thin, thick, dlog, numtau, flux_scale=1e3, nsrc=1000, histogram=True):
    """The counterpart of the headline's quiet medium (VERDICT r3 #5a): BASELINE configs[3] (log-normal density, sources on the
    densest cells) with fluxes x `flux_scale`, so that one converged time step of 1 Myr ionises ~20 % of the volume; the SECOND
    time step then starts with fronts everywhere.  Per outer iteration of that step: duration of the raytrace and of the fused
    pass (HIP events), count of non-converged cells; for its first iteration the histogram of do_chemistry trip counts.
    The library state (sources, grids) is replaced: call last."""
    from pyc2ray_amd.utils.sourceutils import format_sources
    ndens, xh, temp, dr, pos, flux = make_workload("cosmo", N, nsrc)
    p0, f0 = format_sources(pos, flux * flux_scale)
    lib.source_data_to_device(p0, f0, nsrc)
    lib.grid_to_device(_capi.GRID_NDENS, ndens)
    lib.grid_to_device(_capi.GRID_TEMP, temp)
    lib.grid_to_device(_capi.GRID_XH, xh)
    chem = (MYR, BH00, ALBPOW, COLH0, TEMPH0, ABU_C)
    conv_fraction = 1e-4
    conv_criterion = min(int(conv_fraction * N ** 3), (nsrc - 1) / 3)
    lib.set_option(_capi.OPT_TIMING, 1)

    def time_step(want_hist):
        lib.evolve_begin(*chem, R, SIG, dr, MINLOGTAU, dlog, numtau, 0, nsrc, conv_criterion, conv_fraction)
        rows, hist, done = [], None, False
        while not done and len(rows) < 200:
            x_before = lib.grid_to_host(_capi.GRID_XH, np.empty((N, N, N))) if (want_hist and not rows) else None
            lib.kernel_time_reset()
            lib.evolve_enqueue(1)
            _, done, r = lib.evolve_poll(4)
            ch_ms, _ = lib.kernel_time_ms(_capi.KERNEL_CHEMISTRY)
            rt_ms, _ = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)
            rows.append((rt_ms, ch_ms, int(r[0][0])))
            if x_before is not None:
                g = lib.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
                nit = trip_counts(chem[0], ndens, temp, x_before, x_before, g, *chem[1:])
                c = np.bincount(nit.ravel(), minlength=4)
                hist = {"trip_count_cells": {str(k): int(v) for k, v in enumerate(c) if v}, "mean_trip_count": float(nit.mean()),
                        "max_trip_count": int(nit.max()), "mean_of_wave_maxima": float(nit.reshape(-1, 64).max(axis=1).mean())}
        return rows, hist

    rows1, _ = time_step(False)
    x1 = lib.grid_to_host(_capi.GRID_XH_INTERMED, np.empty((N, N, N)))
    lib.grid_to_device(_capi.GRID_XH, x1)
    rows2, hist = time_step(histogram)
    lib.set_option(_capi.OPT_TIMING, 0)
    rt = [r[0] for r in rows2]
    ch = [r[1] for r in rows2]
    # the same accounting as the headline's roofline: 32 B per rate-receiving pair, 16 B for a pair whose exactly-zero rate was
    # not added
    # (summed over the launches since evolve_begin: per launch)
    gamma_cells = lib.last_raytrace_counts()[0] // max(len(rows2), 1)
    zero_cells = lib.last_raytrace_zero_rates() // max(len(rows2), 1)
    rt_bytes = RT_BYTES_PER_UPDATE * (gamma_cells - zero_cells) + (RT_BYTES_PER_UPDATE - 16) * zero_cells
    rt_GBs = rt_bytes / (float(np.mean(rt)) * 1e-3) / 1e9
    ch_GBs = CHEM_BYTES_PER_UPDATE * N ** 3 / (float(np.mean(ch)) * 1e-3) / 1e9
    return {
        "roofline_frac": rt_GBs / HBM_PEAK_GBS, "raytrace_achieved_GBs": rt_GBs,
        "fused_pass_roofline_frac": ch_GBs / HBM_PEAK_GBS,
        "raytrace_pairs_per_launch": int(gamma_cells), "exact_zero_rates_not_added": int(zero_cells), "numtau": int(numtau),
        "workload": f"BASELINE configs[3] ({N}^3 log-normal, {nsrc} sources on the densest cells, r_RT={R:g}), fluxes x {flux_scale:g}, "
                    "dt = 1 Myr: the SECOND time step, which starts with the fronts of the first",
        "ionised_volume_fraction_at_start": float((x1 > 0.5).mean()), "mean_x_at_start": float(x1.mean()),
        "outer_iterations_step_1": len(rows1), "outer_iterations": len(rows2),
        "raytrace_ms": [round(v, 4) for v in rt], "fused_pass_ms": [round(v, 4) for v in ch],
        "raytrace_ms_mean": float(np.mean(rt)), "fused_pass_ms_mean": float(np.mean(ch)), "fused_pass_ms_max": float(np.max(ch)),
        "nonconverged": [r[2] for r in rows2],
        "first_iteration_trip_counts": hist,
        "note": "the headline's medium is the CHEAPEST state of both kernels (converged field: every cell leaves do_chemistry after one "
                "trip; optical depths inside the table); this is the same code on an evolving field",
    }


def configs4_block(comm, lib, p, _capi, rank, world, device_id, numtau_arg, want_slab, grid=512, nsrc=100000, R=32.0, iterations=2):
    """BASELINE configs[4] inside an N-rank run, bounded: `grid`^3 log-normal density, `nsrc` sources on the densest cells IN TOTAL,
    sharded like the main workload (slab exchange, or the full-grid all-reduce if that is what the main pass took), r_RT = 32:
    one warm-up iteration + `iterations` timed ones of the device-resident loop over the ranks (TorchComm.slab_enqueue, one poll
    at the end), MAX over the ranks, with `phases_ms`; and rank 0 ALONE on the whole source list through the one-GPU loop (one
    warm-up + one timed iteration) -> `speedup_vs_one_gpu`.  This is the regime where sharding by source pays (one GPU
    needs ~105 ms per iteration, an exchange ~1 ms), next to configs[3] where it cannot.  A COLLECTIVE: every rank calls it.
    The library is re-initialised for `grid`; returns a dict on every rank (rank 0's is printed)."""
    import torch
    import torch.distributed as dist
    from pyc2ray_amd.dist import SlabPlan
    from pyc2ray_amd.utils.sourceutils import format_sources
    t_all = time.perf_counter()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    N = grid
    if p.cuda_is_init():
        p.device_close()
    p.device_init(N, 64, device_id=device_id)
    thin, thick, dlog = make_tables(numtau_arg)
    p.photo_table_to_device(thin, thick)
    numtau = thin.shape[0] - 1
    ndens, xh, temp, dr, pos, flux = make_workload("cosmo", N, nsrc)
    t_workload = time.perf_counter() - t_all
    lib.grid_to_device(_capi.GRID_NDENS, ndens)
    lib.grid_to_device(_capi.GRID_TEMP, temp)
    lib.grid_to_device(_capi.GRID_XH, xh)
    del ndens, temp, xh
    chem = (MYR, BH00, ALBPOW, COLH0, TEMPH0, ABU_C)

    def fence():
        lib.synchronize()
        if dev == "cuda":
            torch.cuda.synchronize()
        comm.Barrier()

    # rank 0 alone, the whole list, one-GPU loop: the denominator
    one_gpu_ms = None
    if rank == 0:
        try:
            pa, fa = format_sources(pos, flux)
            lib.source_data_to_device(pa, fa, flux.shape[0])
            lib.evolve_begin(*chem, R, SIG, dr, MINLOGTAU, dlog, numtau, 0, flux.shape[0], -1.0, 0.0)
            lib.evolve_enqueue(1); lib.evolve_poll(1); lib.synchronize()
            t0 = time.perf_counter()
            lib.evolve_enqueue(1); lib.evolve_poll(1); lib.synchronize()
            one_gpu_ms = (time.perf_counter() - t0) * 1e3
        except Exception as e:
            print(f"bench: configs4 one-GPU reference failed: {type(e).__name__}: {e}", file=sys.stderr)
    comm.Barrier()

    spos, sflux, bounds = comm.shard_sources_by_slab(pos, flux, world)
    plan = SlabPlan(N, world, R, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(world)])
    lo, hi = bounds[rank], bounds[rank + 1]
    p0, f0 = format_sources(spos[:, lo:hi], sflux[lo:hi])
    lib.source_data_to_device(p0, f0, hi - lo)
    comm.exchange = "slab" if want_slab else "allreduce"
    if want_slab:
        comm.slab_begin(lib, plan, N, R, SIG, dr, hi - lo, MINLOGTAU, dlog, numtau, chem, -1.0, 0.0)
    else:
        comm.reduce_begin(lib, N, R, SIG, dr, hi - lo, MINLOGTAU, dlog, numtau, chem, -1.0, 0.0)
    comm.slab_enqueue(lib, 1)                  # warm-up: geometry tables, first touch, the first exchange
    comm.slab_poll(lib, 1)
    comm.phase_reset()
    comm.phase_timing = True
    fence()
    t0 = time.perf_counter()
    for _ in range(iterations):
        comm.slab_enqueue(lib, 1)
    comm.slab_poll(lib, iterations)
    fence()
    elapsed = time.perf_counter() - t0
    comm.phase_timing = False
    phases = None
    try:
        phases = comm.phase_report()
    except Exception as e:
        print(f"bench: configs4 phase report failed: {type(e).__name__}: {e}", file=sys.stderr)
    gamma, _ = lib.last_raytrace_counts()
    n_done, _, _ = comm.slab_poll(lib, 0)
    t = torch.tensor([elapsed, float(gamma // max(n_done, 1))], dtype=torch.float64, device=dev)
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    ms_step = float(tmax[0].item()) / iterations * 1e3
    pairs = int(round(float(tsum[1].item())))
    return {
        "workload": f"BASELINE configs[4], bounded: {N}^3 log-normal density, {nsrc} sources on the densest cells in total sharded over {world} ranks, "
                    f"r_RT={R:g}, {iterations} outer iterations of the device-resident loop over the ranks after one warm-up iteration",
        "exchange": "slab" if want_slab else "allreduce", "ms_per_step": ms_step, "steps": iterations,
        "value": (pairs + N ** 3) / (ms_step * 1e-3), "unit": "cell-updates/s", "raytrace_updates_per_step": pairs,
        "sources_rank0": hi - lo, "phases_ms": phases,
        "one_gpu_same_workload_ms_per_step": one_gpu_ms, "speedup_vs_one_gpu": (one_gpu_ms / ms_step) if one_gpu_ms else None,
        "busiest_link_bytes_per_exchange": plan.largest_transfer() if want_slab else None,
        "workload_build_s": t_workload, "block_wall_s": time.perf_counter() - t_all,
        "note": "every rank builds the workload itself (FFT of the grid, a full sort for the densest cells: workload_build_s on the host); "
                "one_gpu_same_workload = rank 0 alone, one timed iteration of the one-GPU loop on all sources",
    }


def host_topology():
    """What the host offers this process: logical CPUs and sockets of the machine (/proc/cpuinfo), physical cores per socket,
    the CPUs this process may run on (affinity) and the CPU share its control group grants (cgroup quota), whichever is
    tighter -- a container on a big host sees all of the host's CPUs in /proc/cpuinfo but is throttled to its share."""
    logical, sockets, cores = 0, {}, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
                phys = core = None
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
                sockets[phys] = sockets.get(phys, 0) + 1
            elif line.startswith("core id") and phys is not None:
                cores.add((phys, int(line.split(":")[1])))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    n_sock = max(1, len(sockets))
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    try:                                   # cgroup v2, then v1
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    share = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    phys_per_socket = (len(cores) // n_sock) if cores else max(1, logical // n_sock)
    return {"logical_cpus": logical, "sockets": n_sock, "logical_cpus_per_socket": max(1, logical // n_sock),
            "physical_cores_per_socket": max(1, phys_per_socket), "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "usable_cpus": share}


def self_launch(gpus, argv, timeout_s):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start
    `python -m torch.distributed.run --standalone --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process --
    before this process has imported anything that touches the GPU, and never by replacing this process --, pass rank 0's
    single JSON line through to stdout, and exit with the child's return code.  A child that fails or prints no JSON line
    gives a non-zero exit with the tail of its stderr; a child that outlives `timeout_s` is killed with its ranks
    (each of which is a session of its own) and the exit code is 124; so are they when this process is told to end.  (pyc2ray's own multi-rank test needs an external mpirun,
    test/unit_tests_hackathon/4_multiple_sources_mpi/run_test.py:30-34; this one does not.)"""
    import signal
    import socket
    import subprocess
    import threading
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, PYC2RAY_AMD_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench: --gpus %d without a launcher: starting %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env,
                             start_new_session=True)
    tail, lines = [], []

    def descendants(pid):
        """pids of the processes below `pid` (torch.distributed.run puts every rank into a session of its own: a signal to
        the launcher's process group does not reach them, and once the launcher is gone they cannot be found through it)."""
        found, frontier = [], [pid]
        while frontier:
            parent = frontier.pop()
            try:
                with open(f"/proc/{parent}/task/{parent}/children") as f:
                    kids = [int(k) for k in f.read().split()]
            except (OSError, ValueError):
                kids = []
            found += kids
            frontier += kids
        return found

    def kill_ranks():
        ranks = descendants(child.pid)
        try:
            os.killpg(child.pid, signal.SIGTERM)          # the elastic agent ends its workers when asked to
        except OSError:
            pass
        try:
            child.wait(timeout=5)
        except subprocess.TimeoutExpired:
            pass
        for pid in ranks + descendants(child.pid):
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except OSError:
            pass

    def forward(signum, _frame):        # whoever started THIS process gives up: the ranks must not outlive it
        kill_ranks()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, forward)

    def pump_err():
        for line in child.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            tail.append(line)
            del tail[:-60]

    def pump_out():
        for line in child.stdout:
            lines.append(line)
    threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
    for t in threads:
        t.start()
    try:
        rc = child.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        kill_ranks()
        child.wait()
        print(f"bench: the {gpus}-rank child did not finish within {timeout_s:g} s and was killed", file=sys.stderr, flush=True)
        return 124
    for t in threads:
        t.join(timeout=10)
    found = None
    for line in lines:
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if isinstance(d, dict) and ("metric" in d or "launch_check" in d):
            found = line.strip()
    if rc != 0:
        print(f"bench: the {gpus}-rank child exited with code {rc}; last lines of its stderr:\n" + "".join(tail[-25:]),
              file=sys.stderr, flush=True)
        return rc
    if found is None:
        print("bench: the child exited with code 0 but printed no JSON line; its stdout was:\n" + "".join(lines[-20:]),
              file=sys.stderr, flush=True)
        return 1
    print(found, flush=True)
    return 0


def launch_check(fail_rank):
    """`--launch-check [RANK]`: every rank brings the process group up (the backend bench.py would use), sums one number over
    the ranks and rank 0 prints one JSON line -- the launch path of an N-rank run without its measurement (CPU test of the
    self-launch: tests/test_dist_gloo.py).  With RANK >= 0 that rank raises instead, to show how a failing rank ends the run."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("PYC2RAY_AMD_BENCH_BACKEND", "nccl")
    if rank == fail_rank:
        raise SystemExit(f"bench --launch-check: rank {rank} fails on request")
    if os.environ.get("PYC2RAY_AMD_BENCH_TEST_HANG") == "1":        # (tests: a child that never finishes)
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    from pyc2ray_amd.dist import init_process_group_from_env
    os.environ.setdefault("PYC2RAY_AMD_DIST_TIMEOUT_S", "60")
    init_process_group_from_env(backend)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    ok = float(t.item()) == world * (world + 1) / 2.0
    if rank == 0:
        print(json.dumps({"launch_check": bool(ok), "world_size_reported_by_backend": dist.get_world_size(), "backend": backend,
                          "self_launched": os.environ.get("PYC2RAY_AMD_BENCH_SELF_LAUNCHED", "0") == "1",
                          "visible_gpus": torch.cuda.device_count(), "host": host_topology()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    if "--cpu-worker" in sys.argv:
        return _cpu_worker()
    ap = argparse.ArgumentParser()
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 started without a launcher: seconds the self-launched N-rank child may take")
    ap.add_argument("--launch-check", type=int, nargs="?", const=-1, default=None, metavar="FAILING_RANK",
                    help="only bring the ranks up, sum one number over them and print one JSON line (the launch path without the "
                         "measurement); with a rank number that rank fails on purpose")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of --steps steps is run this many times back to back; value / ms_per_step are the "
                         "MEDIAN region, `spread` holds the fastest and the slowest")
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--nsrc", type=int, default=1000, help="sources: in total (strong scaling) or per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = --nsrc sources in total sharded over the ranks (the metric), weak = --nsrc per rank")
    ap.add_argument("--exchange", choices=["auto", "slab", "allreduce"], default="auto",
                    help="N>1: how the per-rank rates are summed (see the module docstring); auto = whichever the byte model "
                         "of pyc2ray_amd/dist.py makes cheaper for this (ranks, mesh, radius, sources), printed in config")
    ap.add_argument("--slab-chunks", type=int, default=0,
                    help="N>1, slab exchange: trace chunks per step; planes final after a chunk travel while the next is traced "
                         "(1 = no overlap; 0 = TorchComm's default: 2 with two ranks, else 1)")
    ap.add_argument("--R", type=float, default=32.0)
    ap.add_argument("--numtau", type=int, default=NUMTAU,
                    help="entries of the rate tables: 20000 = the benchmark's parameter file (default, BASELINE configs[2]); 2000 = the "
                         "reference's production parameter files")
    ap.add_argument("--workload", choices=["uniform", "cosmo"], default=None,
                    help="default: uniform (configs[2]) on one GPU, cosmo (configs[3]) on several")
    ap.add_argument("--cpu-sources", type=int, default=512,
                    help="sources in the single-core CPU-baseline sample (0 = skip): 512 = ~11 s of the reference Fortran + 1 s for its pass")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="worker processes of the one-socket CPU figure (0 = the physical cores of one socket, as far as this "
                         "process may use them; 1 = skip it)")
    ap.add_argument("--z-transposed", type=int, default=1)
    ap.add_argument("--block-threads", type=int, default=0, help="raytrace workgroup size (0 = auto)")
    ap.add_argument("--overlap", type=int, default=-1,
                    help="N>1: 1 = pipeline the all-reduce of the rate grid with the raytrace (sources traced in order of "
                         "their first coordinate), 0 = trace, then all-reduce; default: env PYC2RAY_AMD_OVERLAP or 0")
    ap.add_argument("--sectors", type=int, default=0, help="decomposition of a source (ASORA_OPT_SECTORS): 0 auto, 1 octants, 2 sectors, 3 sector pairs, 5 octant pairs, 6 whole sphere, 7 half spheres, 8 all-sign sectors")
    ap.add_argument("--pair-sources", type=int, default=0, help="raytrace, two sources per workgroup: 0 auto, 1 never, 2 always")
    ap.add_argument("--one-gpu-reference", type=int, default=1,
                    help="N>1: before the multi-rank region rank 0 ALONE runs the same workload (all sources) through the one-GPU loop "
                         "while the others wait -> one_gpu_same_workload_ms_per_step, speedup_vs_one_gpu (0 = skip)")
    ap.add_argument("--configs4", type=int, default=-1,
                    help="N>1: after the metric's workload also a bounded pass of BASELINE configs[4] (512^3, 1e5 sources, the regime where "
                         "sharding pays) -> `configs4`; -1 = on for the default N-rank job, 0 = off, 1 = on")
    ap.add_argument("--configs4-grid", type=int, default=512)
    ap.add_argument("--configs4-nsrc", type=int, default=100000)
    ap.add_argument("--evolving-state", type=int, default=-1,
                    help="one GPU: also time the trace and the fused pass on a grid WITH ionisation fronts (configs[3], fluxes x 1e3, "
                         "second time step) -> `evolving_state`; default: on for the default job only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: be the launcher (nothing has touched the GPU yet; the N ranks are children)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.launch_check is not None:
        return launch_check(args.launch_check)
    N, K, W = args.N, args.steps, args.warmup
    if args.workload is None:
        args.workload = "uniform" if world == 1 else "cosmo"
    strong = world > 1 and args.scaling == "strong"
    nsrc_total = args.nsrc if (strong or world == 1) else args.nsrc * world

    # the one-socket CPU sample runs in worker processes that are started now, before this process touches the GPU,
    # and sit idle until the GPU measurement is over.  Worker count: the physical cores of ONE socket of this host
    # (north_star: "single-socket Fortran CPU"), or fewer when this process may not use that many CPUs (affinity, cgroup
    # quota) or when the host's free memory does not hold that many copies of the reference routine's three output grids
    # -- the figure for the whole socket is then extrapolated linearly and labelled so.
    cpu_workers, cpu_cores, cpu_dir = None, 1, None
    tables = None
    workload = None
    topo = host_topology()
    cpu_note = None
    if world == 1 and args.cpu_sources > 0 and args.cpu_cores != 1:
        try:
            socket_cores = topo["physical_cores_per_socket"]
            cpu_cores = args.cpu_cores if args.cpu_cores > 1 else min(socket_cores, topo["usable_cpus"])
            per_worker = 3.3 * 8.0 * N ** 3 + 3.0e8            # phi_ion, coldensh_out, touched part of phi_heat + the interpreter
            room = usable_memory_bytes()
            if room is not None and cpu_cores * per_worker > 0.5 * room:
                capped = max(1, int(0.5 * room / per_worker))
                cpu_note = f"{cpu_cores} workers wanted, {capped} fit half of the {room / 2**30:.0f} GiB this process may use"
                cpu_cores = capped
            if cpu_cores > 1:
                tables = make_tables(args.numtau)
                workload = make_workload(args.workload, N, nsrc_total)
                cpu_workers, cpu_dir = start_cpu_workers(cpu_cores, workload, N, args.R,
                                                         min(nsrc_total, max(args.cpu_sources, 16 * cpu_cores)), tables)
        except Exception as e:
            print(f"bench: one-socket CPU sample not started: {type(e).__name__}: {e}", file=sys.stderr)
            cpu_workers = None

    import pyc2ray_amd as p
    from pyc2ray_amd import _capi
    from pyc2ray_amd.load_extensions import load_asora
    from pyc2ray_amd.utils.sourceutils import format_sources

    comm = None
    links = None
    saved_stdout = None
    if world > 1 or os.environ.get("PYC2RAY_AMD_FORCE_COLLECTIVE", "0") == "1":
        # RCCL prints a version banner on stdout when its first communicator comes up; the contract is ONE JSON
        # line on stdout, so everything before that line is sent to stderr (at the file-descriptor level)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch
        import torch.distributed as dist
        from pyc2ray_amd.dist import TorchComm, init_process_group_from_env
        # (rehearsal of the multi-rank logic on a box with ONE GPU: PYC2RAY_AMD_BENCH_BACKEND=gloo puts every rank on
        #  device PYC2RAY_AMD_BENCH_DEVICE and stages the sums through the host; never the measured configuration)
        backend = os.environ.get("PYC2RAY_AMD_BENCH_BACKEND", "nccl")
        os.environ.setdefault("PYC2RAY_AMD_DIST_TIMEOUT_S", "180")     # a collective that never completes ends the run after 3 minutes
        init_process_group_from_env(backend)
        comm = TorchComm()

        def all_ranks_ok(ok):
            """MIN over the ranks of a yes/no: every rank takes the same branch afterwards."""
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return float(flag.item()) != 0.0

        # Gate 1, before anything else uses the point-to-point path: the slab exchange has never run between real GPUs on the
        # build box (it has one).  One small ring of sends and receives through the same torch call; any rank that fails or
        # receives a wrong payload sends EVERY rank to the all-reduce path (MIN vote).
        p2p_ok = True
        try:
            p2p_ok = comm.preflight_p2p(N * N)
        except Exception as e:
            print(f"bench: point-to-point preflight failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
            p2p_ok = False
        p2p_ok = all_ranks_ok(p2p_ok)
        # What the links of THIS job deliver, through the calls the two exchange schemes use (16 MiB point-to-point ring -- only
        # if the preflight went through --, all-reduce of one N^3 grid): `--exchange auto` decides from these, not from assumed
        # rates.  A failure on ANY rank makes every rank drop the measurement (vote), and rank 0's numbers are broadcast, so
        # every rank derives the same exchange choice from the same numbers.
        links = None
        try:
            links = comm.measure_links(p2p_bytes=16 << 20, allreduce_bytes=8 * N ** 3, p2p=p2p_ok)
        except Exception as e:
            print(f"bench: link measurement failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
        if not all_ranks_ok(links is not None):
            links = None
        else:
            box = [links]
            dist.broadcast_object_list(box, src=0)
            links = box[0]

    visible_gpus, backend_world = None, None
    try:
        import torch as _t
        visible_gpus = _t.cuda.device_count()
        if comm is not None:
            backend_world = dist.get_world_size()
    except Exception:
        pass
    lib = load_asora()
    p.device_init(N, 64, device_id=int(os.environ.get("PYC2RAY_AMD_BENCH_DEVICE", local_rank)))
    thin, thick, dlog = tables if tables is not None else make_tables(args.numtau)
    p.photo_table_to_device(thin, thick)
    numtau = thin.shape[0] - 1                  # as raytracing_benchmark/run_test.py:85 passes it

    ndens, xh, temp, dr, pos, flux = workload if workload is not None else make_workload(args.workload, N, nsrc_total)
    overlap = comm is not None and (args.overlap == 1 or (args.overlap < 0 and comm.overlap))
    slab = comm is not None and args.exchange in ("slab", "auto") and not overlap
    fell_back = None
    if slab and not p2p_ok:
        slab, fell_back = False, "point-to-point preflight"
    exchange_model = None
    src_i0 = None
    plan = None
    if slab:
        # contiguous blocks of the list ordered by first coordinate: what each rank exchanges is then a few planes
        from pyc2ray_amd.dist import SlabPlan
        comm.exchange = "slab"
        pos, flux, bounds = comm.shard_sources_by_slab(pos, flux, world)
        if not strong:                       # weak scaling: exactly --nsrc per rank
            bounds = [r * args.nsrc for r in range(world + 1)]
        lo, hi = bounds[rank], bounds[rank + 1]
        plan = SlabPlan(N, world, args.R, [pos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(world)])
        if args.slab_chunks > 0:
            comm.slab_chunks = args.slab_chunks
        # what each scheme moves per step over its busiest link (exact: SlabPlan) and what that costs at the rates MEASURED in this
        # job a moment ago (`links`); ring all-reduce: 2 (P-1) steps of N^3/P doubles over one link each -- its time for one
        # N^3 grid was measured directly
        Kc = plan.common_chunks(comm.slab_chunks)
        link = plan.largest_transfer()
        sched = [plan.send_schedule(r, Kc) for r in range(world)]
        early = [sum((b - a) for pieces in sc[:-1] for _, a, b in pieces) for sc in sched]
        total = [sum((b - a) for pieces in sc for _, a, b in pieces) for sc in sched]
        hidden = min((e / t) if t else 1.0 for e, t in zip(early, total))          # share of exchange 1 that leaves before the last chunk
        ring = 2.0 * (world - 1) / world * 8.0 * N ** 3
        measured = links is not None and links.get("p2p_GBs") and links.get("allreduce_ms")
        exchange_model = {
            "slab_bytes_per_link_and_exchange": link, "slab_exchanges_per_step": 2, "slab_trace_chunks": Kc,
            "slab_share_of_rate_exchange_sent_before_the_last_chunk": hidden,
            "allreduce_ring_bytes_per_link_per_step": ring,
            "rates_from": "measured in this job (measured_link_GBs)" if measured else "ASSUMED 50 GB/s per link (the link measurement failed)",
        }
        p2p_gbs = links["p2p_GBs"] if measured else 50.0
        slab_ms = 2.0 * link / (p2p_gbs * 1e6)
        allreduce_ms = links["allreduce_ms"] if measured else ring / (50.0 * 1e6)
        exchange_model.update({
            "slab_comm_ms": slab_ms, "slab_comm_ms_with_overlap": (2.0 - hidden) * link / (p2p_gbs * 1e6),
            "allreduce_comm_ms": allreduce_ms,
            "note": "bytes are exact (SlabPlan); slab_comm_ms = two exchanges of the busiest link's bytes at the measured point-to-point "
                    "rate, allreduce_comm_ms = the measured all-reduce of one N^3 grid; the slab scheme also runs 1/P of the chemistry per rank",
        })
        if args.exchange == "auto" and slab_ms > allreduce_ms:
            slab = False                         # (every rank reaches nearly every plane, or the point-to-point path is slow here)
        exchange_model["choice"] = "slab" if slab else "allreduce"
    if comm is not None and not slab:
        comm.exchange = "allreduce"
    if slab:
        pass
    elif strong:
        per = nsrc_total // world                                    # evolve.py:362-367
        lo, hi = rank * per, ((rank + 1) * per if rank != world - 1 else nsrc_total)
    else:
        lo, hi = rank * args.nsrc, (rank + 1) * args.nsrc          # contiguous block per rank (evolve.py:362-367)
    my_pos, my_flux = pos[:, lo:hi], flux[lo:hi]
    n_local = hi - lo
    if overlap:
        comm.overlap = True
        my_pos, my_flux = comm.sort_sources_for_overlap(my_pos, my_flux)
        src_i0 = my_pos[0].astype(np.int64) - 1
    elif comm is not None:
        comm.overlap = False
    p0, f0 = format_sources(my_pos, my_flux)
    lib.grid_to_device(_capi.GRID_NDENS, ndens)
    lib.grid_to_device(_capi.GRID_TEMP, temp)
    lib.grid_to_device(_capi.GRID_XH, xh)
    lib.set_option(_capi.OPT_Z_TRANSPOSED, args.z_transposed)
    lib.set_option(_capi.OPT_BLOCK_THREADS, args.block_threads)
    lib.set_option(_capi.OPT_SECTORS, args.sectors)
    lib.set_option(_capi.OPT_PAIR_SOURCES, args.pair_sources)

    chem = (MYR, BH00, ALBPOW, COLH0, TEMPH0, ABU_C)
    from pyc2ray_amd.evolve import EVOLVE_BATCH
    poll_every = max(1, min(EVOLVE_BATCH, 32))      # evolve3D reads the device's status back once per batch of iterations

    def one_gpu_region(n_sources, steps):
        """`steps` outer iterations of the ONE-GPU loop exactly as evolve3D drives it (evolve.py:_device_loop): enqueue, and
        after every batch of EVOLVE_BATCH iterations one asora_evolve_poll -- the 24-byte status read-back and the fold of the
        last iteration's rate accumulators into phi_ion.  Returns the last poll's history rows."""
        rows = []
        for s_ in range(steps):
            lib.evolve_enqueue(1)          # raytrace + fused chemistry pass + convergence test
            if (s_ + 1) % poll_every == 0 or s_ == steps - 1:
                _, _, r_ = lib.evolve_poll(poll_every)
                rows = r_ if len(r_) else rows
        return rows

    # N>1: the SAME workload -- the whole source list on the same density -- on ONE GPU, inside this job: rank 0 alone, through
    # the one-GPU loop, while the other ranks wait at a barrier.  The denominator of `speedup_vs_one_gpu` (a `--gpus 1` run
    # of this script measures configs[2], another workload).
    one_gpu_ms, one_gpu_how = None, None
    if comm is not None and args.one_gpu_reference:
        if rank == 0:
            try:
                # bounded: the other ranks wait at a barrier whose timeout is PYC2RAY_AMD_DIST_TIMEOUT_S.  One iteration is timed
                # first; the full protocol (warm-up + `repeats` regions of K steps) runs only if it fits a fifth of that
                # timeout, else one region of min(K, 3) steps, else that single iteration is the figure.
                budget = 0.2 * float(os.environ.get("PYC2RAY_AMD_DIST_TIMEOUT_S", "180"))
                pa, fa = format_sources(pos, flux)
                lib.source_data_to_device(pa, fa, flux.shape[0])
                lib.evolve_begin(*chem, args.R, SIG, dr, MINLOGTAU, dlog, numtau, 0, flux.shape[0], -1.0, 0.0)
                one_gpu_region(flux.shape[0], 1)
                lib.synchronize()
                t0 = time.perf_counter()
                one_gpu_region(flux.shape[0], 1)
                lib.synchronize()
                t_one = time.perf_counter() - t0
                reps_ = max(1, args.repeats)
                if t_one * (W + K * reps_) <= budget:
                    plan_, how = (W, K, reps_), f"{reps_} regions of {K} steps after {W} warm-up steps, median"
                elif t_one * min(K, 3) <= budget:
                    plan_, how = (0, min(K, 3), 1), f"one region of {min(K, 3)} steps (the full protocol would not fit {budget:.0f} s)"
                else:
                    plan_, how = None, "a single iteration (one iteration takes %.1f s)" % t_one
                if plan_ is None:
                    one_gpu_ms = t_one * 1e3
                else:
                    one_gpu_region(flux.shape[0], plan_[0]) if plan_[0] else None
                    regs = []
                    for _ in range(plan_[2]):
                        lib.synchronize()
                        t0 = time.perf_counter()
                        one_gpu_region(flux.shape[0], plan_[1])
                        lib.synchronize()
                        regs.append((time.perf_counter() - t0) / plan_[1])
                    one_gpu_ms = float(np.median(regs)) * 1e3
                one_gpu_how = how
            except Exception as e:
                print(f"bench: one-GPU reference run failed: {type(e).__name__}: {e}", file=sys.stderr)
        comm.Barrier()
    lib.source_data_to_device(p0, f0, n_local)
    # "loop": the multi-rank step runs on the device-resident loop -- slab exchange, or the full-grid all-reduce when it is not pipelined
    state = {"slab": slab, "loop": comm is not None and (slab or (not overlap and getattr(comm, "device_loop", False))), "unpolled": 0, "rows": [], "host_enqueue_s": 0.0,
             "host_enqueued": 0}

    def begin_time_step():
        # conv_criterion = -1 and convergence_fraction = 0 can never be met: every enqueued iteration does its work
        if comm is None:
            lib.evolve_begin(*chem, args.R, SIG, dr, MINLOGTAU, dlog, numtau, 0, n_local, -1.0, 0.0)
        elif state["slab"]:
            poll_slab()
            comm.slab_begin(lib, plan, N, args.R, SIG, dr, n_local, MINLOGTAU, dlog, numtau, chem, -1.0, 0.0)
        elif state["loop"]:
            poll_slab()
            comm.reduce_begin(lib, N, args.R, SIG, dr, n_local, MINLOGTAU, dlog, numtau, chem, -1.0, 0.0)
        else:
            lib.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)              # evolve.py:136-137
            lib.grid_copy(_capi.GRID_XH_INTERMED, _capi.GRID_XH)

    def poll_slab():
        """What evolve3D_MPI does once per batch of iterations: status read-back + fold of the own rates (asora_evolve_poll)."""
        if state["unpolled"]:
            _, _, r_ = comm.slab_poll(lib, poll_every)
            state["unpolled"] = 0
            if len(r_):
                state["rows"] = r_

    def step():
        if state["loop"]:              # what evolve3D_MPI does per outer iteration with a TorchComm: the device loop over the ranks
            t_ = time.perf_counter()
            comm.slab_enqueue(lib, 1)
            state["host_enqueue_s"] += time.perf_counter() - t_       # host time to ISSUE an iteration (nothing in it waits for the GPU)
            state["host_enqueued"] += 1
            state["unpolled"] += 1
            if state["unpolled"] >= poll_every:
                poll_slab()
            return (state["rows"][-1][:3] if len(state["rows"]) else None)
        if comm is not None:           # full-grid all-reduce (optionally pipelined), chemistry on every rank
            return comm.raytrace_and_allreduce(lib, N, args.R, SIG, dr, n_local, MINLOGTAU, dlog, numtau,
                                               src_i0=src_i0, chemistry=chem)
        lib.evolve_enqueue(1)          # one outer iteration of evolve3D: raytrace + fused chemistry pass + convergence test
        return None

    def fence():
        if comm is not None and state["loop"]:
            poll_slab()                # a region ends with the poll of its last batch, as on one GPU
        lib.synchronize()
        if comm is not None:
            import torch
            torch.cuda.synchronize()
            comm.Barrier()
            torch.cuda.synchronize()

    if comm is not None and state["slab"]:
        # Gate 2: the first full slab step, followed by the same vote.  If it raises on ANY rank, every rank takes the full-grid
        # all-reduce (any partition of the sources is fine for it) instead of losing the run; which path ran, and why, is in
        # the JSON (`config.parallelism`, `config.exchange_fallback`).  Nothing is re-executed: the process keeps its GPU context.
        ok = True
        try:
            begin_time_step(); step(); fence()
        except Exception as e:
            print(f"bench: first slab step failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
            ok = False
        if not all_ranks_ok(ok):
            state["slab"] = False
            state["unpolled"] = 0
            slab = False
            comm.exchange = "allreduce"
            fell_back = "first slab step"
            if exchange_model is not None:
                exchange_model["choice"] = "allreduce (fallback)"
    # the FIRST iteration of a time step additionally forms nHI from xh and zeroes the accumulators on the whole grid
    begin_time_step(); step(); fence()
    t0 = time.perf_counter()
    begin_time_step(); step(); fence()
    first_iteration_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(W):
        step()
    if comm is None:
        lib.evolve_poll(0)
    lib.set_option(_capi.OPT_TIMING, 1)
    lib.kernel_time_reset()
    if comm is not None:
        comm.phase_reset()
        comm.phase_timing = True       # where a step's time goes (HIP events on the library's stream; `phases_ms`)
    # the timed region -- exactly K steps between two fences -- `repeats` times back to back: one region lasts a few tens
    # of milliseconds, and the fused pass alone varies by 20-30 % from box to box and with the box's temperature.
    # One GPU: the polls evolve3D makes (one per EVOLVE_BATCH iterations: status read-back + fold of the rates) are
    # INSIDE the region.
    regions = []
    conv = None
    last_rows = []
    n_timed = 0
    state["host_enqueue_s"], state["host_enqueued"] = 0.0, 0
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        if comm is None:
            rows_ = one_gpu_region(n_local, K)
            last_rows = rows_ if len(rows_) else last_rows
        else:
            for _ in range(K):
                conv = step()
        fence()
        regions.append(time.perf_counter() - t0)
        n_timed += K
    lib.set_option(_capi.OPT_TIMING, 0)
    phases_ms = None
    if comm is not None:
        comm.phase_timing = False
        try:
            phases_ms = comm.phase_report()           # mean per step, MAX over the ranks (a collective)
        except Exception as e:
            print(f"bench: phase report failed: {type(e).__name__}: {e}", file=sys.stderr)

    gamma_cells, eval_cells = lib.last_raytrace_counts()
    zero_cells = lib.last_raytrace_zero_rates()
    raytrace_variant = lib.last_raytrace_variant()
    if comm is not None and state["loop"]:
        # the counters of the device loop over the ranks run on from its begin, like the one-GPU loop's
        n_done, _, _ = comm.slab_poll(lib, 0)
        n_done = max(n_done, 1)
        gamma_cells, eval_cells, zero_cells = gamma_cells // n_done, eval_cells // n_done, zero_cells // n_done
    if slab:
        comm.slab_gather(lib, plan, _capi.GRID_XH_INTERMED, N)
        comm.slab_gather(lib, plan, _capi.GRID_PHI_ION, N)
    if comm is None:
        # the counters of the device-resident loop run on from evolve_begin: per iteration = total / iterations
        n_done, _, rows = lib.evolve_poll(min(K, 32))
        gamma_cells, eval_cells, zero_cells = gamma_cells // n_done, eval_cells // n_done, zero_cells // n_done
        rows = rows if len(rows) else last_rows
        conv = (rows[-1][0],) if len(rows) else (0,)
    rt_ms, rt_n = lib.kernel_time_ms(_capi.KERNEL_RAYTRACE)
    ch_ms, ch_n = lib.kernel_time_ms(_capi.KERNEL_CHEMISTRY)
    pr_ms, pr_n = lib.kernel_time_ms(_capi.KERNEL_PREP)
    fi_ms, fi_n = lib.kernel_time_ms(_capi.KERNEL_FINISH)

    tot_gamma = gamma_cells
    if comm is not None:
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tmax = torch.tensor(regions, dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)            # a region lasts as long as its slowest rank
        regions = [float(v) for v in tmax.cpu().tolist()]
        t = torch.tensor([float(gamma_cells)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot_gamma = int(round(t[0].item()))
    elapsed = float(np.median(regions))
    # after the timed region: every rank must hold the same summed rates and the same chemistry result
    ranks_agree = None
    if comm is not None:
        try:
            phi_h = lib.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
            x_h = lib.grid_to_host(_capi.GRID_XH_INTERMED, np.empty((N, N, N)))
            mine = torch.tensor([float(phi_h.sum()), float(np.abs(phi_h).max()), float(x_h.sum())], dtype=torch.float64,
                                device="cuda" if dist.get_backend() == "nccl" else "cpu")
            lo_, hi_ = mine.clone(), mine.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            ranks_agree = bool(torch.equal(lo_, hi_)) and bool(np.isfinite(phi_h).all()) and float(phi_h.sum()) > 0.0
        except Exception as e:    # reporting only
            print(f"bench: rank-agreement check failed: {type(e).__name__}: {e}", file=sys.stderr)

    grid_placement = lib.debug_placement()         # (of the metric's grids: the configs4 block below re-initialises the library)
    configs4 = None
    if comm is not None and state["loop"] and (args.configs4 == 1 or (args.configs4 < 0 and args.workload == "cosmo" and N == 256 and
                                                                      args.nsrc == 1000 and args.R == 32.0 and strong)):
        # every rank takes the same branch (same arguments, same exchange choice) -- and votes on the outcome, so that a failure
        # on one rank costs this block, not the line
        ok = True
        try:
            configs4 = configs4_block(comm, lib, p, _capi, rank, world, int(os.environ.get("PYC2RAY_AMD_BENCH_DEVICE", local_rank)),
                                      args.numtau, state["slab"], grid=args.configs4_grid, nsrc=args.configs4_nsrc)
        except Exception as e:
            print(f"bench: configs4 block failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
            configs4, ok = {"failed": f"{type(e).__name__}: {e}"}, False
        try:
            if not all_ranks_ok(ok) and ok:
                configs4 = {"failed": "on another rank (see stderr)"}
        except Exception as e:
            print(f"bench: configs4 vote failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
    if rank != 0:
        p.device_close()
        if comm is not None:
            dist.destroy_process_group()
        return

    units_per_step = tot_gamma + N ** 3
    value = units_per_step * K / elapsed
    rt_launch_s = (rt_ms / max(rt_n, 1)) * 1e-3
    # algorithmic bytes of a launch (8 ndens + 8 xh_av + 16 rate read-modify-write per rate-receiving pair).  A pair
    # whose rate is exactly +0 and is therefore NOT added (ASORA_OPT_SKIP_ZERO_RATES) moves no rate bytes: it counts
    # 16 B, not 32.  0 such pairs in the headline workload.
    rt_bytes = RT_BYTES_PER_UPDATE * (gamma_cells - zero_cells) + (RT_BYTES_PER_UPDATE - 16) * zero_cells
    achieved = rt_bytes / rt_launch_s / 1e9 if rt_n else None
    achieved_all32 = RT_BYTES_PER_UPDATE * gamma_cells / rt_launch_s / 1e9 if rt_n else None
    insphere = 4.0 * np.pi * args.R ** 3 / 3.0
    default_job = (args.workload == "uniform" and args.R == 32.0 and N == 256 and args.nsrc == 1000 and world == 1 and args.numtau == NUMTAU)
    # counters: only from a committed summary collected on THIS build of the library (find_pmc_summary)
    build_id = lib.build_id()
    pmc_summary, pmc_missing = find_pmc_summary(build_id) if default_job else (None, "counters are committed for the default job only")
    PMC_SUMMARY = pmc_summary or "(no counter summary of this build)"
    rt_counters = pmc_counters("raytrace_octant_kernel", pmc_summary)
    ch_launch_s = (ch_ms / max(ch_n, 1)) * 1e-3
    chem_cells = N ** 3 if (comm is None or not slab) else (plan.own[0][1] - plan.own[0][0]) * N * N
    ch_achieved = CHEM_BYTES_PER_UPDATE * chem_cells / ch_launch_s / 1e9 if ch_n else None
    fused_pass = comm is None or state["loop"]   # (the device loop over the ranks runs the same fused pass: on the own planes, or -- behind the all-reduce -- on all)
    ch_actual = (CHEM_FUSED_BYTES_PER_UPDATE if fused_pass else CHEM_BYTES_PER_UPDATE) * chem_cells / ch_launch_s / 1e9 if ch_n else None
    comm_bytes = None
    if slab:
        comm_bytes = 2 * max(sum(plan.bytes_per_rank(r)) for r in range(world))     # two exchanges, sent + received
    elif comm is not None:
        comm_bytes = int(2 * 2 * (world - 1) / world * 8 * N ** 3)                   # ring all-reduce, sent + received

    out = {
        "metric": "cell-updates/sec (raytrace+chem)",
        "value": value,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "repeats": len(regions),
        "spread": {"ms_per_step_min": min(regions) / K * 1e3, "ms_per_step_max": max(regions) / K * 1e3,
                   "value_min": (tot_gamma + N ** 3) * K / max(regions), "value_max": (tot_gamma + N ** 3) * K / min(regions),
                   "ms_per_step_all": [r / K * 1e3 for r in regions],
                   "note": "value and ms_per_step are the median of `repeats` timed regions of `steps` steps each"},
        "higher_is_better": True,
        "scaling": "strong" if (strong or world == 1) else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload_label(args.workload, N, nsrc_total, args.R, world, strong),
            "grid": N, "sources_total": nsrc_total, "sources_rank0": n_local, "R_cells": args.R, "numtau": args.numtau,
            "host_cores": topo["logical_cpus"], "host_cpus_this_process_may_use": topo["usable_cpus"],
            "visible_gpus": visible_gpus, "world_size_reported_by_backend": backend_world,
            "launched_by": ("bench.py itself (child torch.distributed.run)" if os.environ.get("PYC2RAY_AMD_BENCH_SELF_LAUNCHED") == "1"
                            else "an external launcher" if "WORLD_SIZE" in os.environ else "plain python"),
            "library_build_id": lib.build_id(),
            # where device_init put the grids: allocations of the whole arena tried (at most 8 by default), probe time (a kernel with
            # the fused pass's stream mix) of the one kept and of the slowest -- placements differ by ~15 % on some boxes, not at all
            # on others (csrc/api.hip choose_arena) -- and what device_init and, inside it, the probe cost on the host clock
            "grid_placement": grid_placement,
            # which form of the raytrace kernel the timed launches took (asora_last_raytrace_variant)
            "raytrace_variant": raytrace_variant,
            "ranks_agree_on_rates_and_ionised_fraction": ranks_agree,
            "parallelism": ("single GPU" if world == 1 else
                            f"sources sharded over {world} ranks by slab of the first coordinate; rates sent plane-wise to the "
                            f"owners of the planes (trace in {plan.common_chunks(comm.slab_chunks)} chunks, final planes sent while the next "
                            "chunk is traced), slab chemistry, xh_av sent back (pyc2ray_amd/dist.py SlabPlan)" if slab else
                            f"sources x{world}, rate-grid all-reduce " + ("pipelined with the trace" if overlap else
                                                                        "after the trace, on the device-resident loop" if state["loop"] else "after the trace")
                            + ", chemistry on every rank"),
            "comm_bytes_per_rank_per_step": comm_bytes,
            "exchange_requested": args.exchange if world > 1 else None,
            "exchange_model": exchange_model,
            "exchange_fallback": fell_back,
            "unit_definition": "rate-receiving (source,cell) pairs (|d|<=R) + N^3 chemistry cells per step",
            "raytrace_updates_per_step": tot_gamma,
            "chemistry_updates_per_step": N ** 3,
            "column_density_evaluations_per_step_rank0": eval_cells,
            # of raytrace_updates_per_step (rank 0): pairs whose rate is exactly +0 -- a thick cell beyond the last table
            # entry -- and is therefore not added to the grid (bit-identical result; 0 for the headline workload, whose
            # largest optical depth, 32 cells x 227, stays inside the table)
            "exact_zero_rates_not_added_per_step_rank0": zero_cells,
            "nonconverged_cells_last_step": int(conv[0]) if conv is not None else None,
            "first_iteration_of_a_time_step_ms": first_iteration_ms,
            "step_definition": ("steady-state outer iteration of evolve3D: raytrace + one fused pass (chemistry, nHI for the next "
                                "trace, next accumulators zeroed) + convergence test on the device; and, once per %d steps as in "
                                "evolve3D's loop, asora_evolve_poll inside the timed region (status read-back + fold of the last "
                                "iteration's rate accumulators into phi_ion)" % poll_every) if comm is None else
                               "one outer iteration of evolve3D_MPI through the calls it makes (TorchComm.slab_enqueue -- the sharded "
                               "device loop: trace, rates to the owners, ONE fused pass on the own planes, xh_av back, convergence test "
                               "on the device; or, with --exchange allreduce, trace, fold, in-place all-reduce of the rate grid, ONE fused "
                               "pass on the whole grid on every rank, test on the device -- with one slab_poll per %d steps inside the "
                               "timed region; with --overlap 1 the pipelined raytrace_and_allreduce): "
                               "see phases_ms" % poll_every,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "raytrace_octant_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": pmc_traffic_bytes("raytrace_octant_kernel", pmc_summary),
            "traffic_source": (PMC_SUMMARY + ": (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch, separate rocprofv3 --pmc "
                               "passes of this command on this workload, collected on the build of the library that ran here "
                               f"(asora_build_id {build_id}); a committed measurement, not collected in this run")
                              if pmc_summary else None,
            "traffic_unavailable": pmc_missing,
            "library_build_id": build_id,
            "algorithmic_bytes_per_launch": rt_bytes,
            "frac_if_every_pair_counted_32B": (achieved_all32 / HBM_PEAK_GBS) if achieved_all32 else None,
            "bytes_accounting": "32 B per rate-receiving pair; 16 B for a pair whose exactly-zero rate is not added "
                                f"({zero_cells} of {gamma_cells} pairs in this launch)",
            "avg_launch_ms": rt_ms / max(rt_n, 1),
            "launches_timed": rt_n,
            # what the launch is bound by, as this build's OWN ablation has it (a diagnostic build of the same kernel,
            # make EXTRA=-DASORA_ENABLE_ABLATION: profiles/r04_ablate_R16_R32.txt, `--R 32 ablate 0/1/3`)
            "binding_resource": ("co-limited three ways within 10 %%, none saturated.  Ablation on this workload (diagnostic build, wrong results by "
                                 "design): 1.107 ms as built, 1.010 ms with the rate atomics removed, 0.993 ms with the rates removed as well -- the "
                                 "floor is the SWEEP ITSELF: %.3g VALU wave-instructions per launch (`valu_issue.frac` of issue at 4 cycles each) "
                                 "run at two waves per SIMD -- the four shell buffers of a workgroup (two sources x two shells, 13.4 KB each) cap a "
                                 "CU at two workgroups of four waves, 128 VGPRs cap a SIMD at four -- so every wave waits out its own chain LDS "
                                 "read -> interpolation -> division -> LDS write -> shell barrier.  Second, the memory-side rate atomics: every double "
                                 "added to the rate grid leaves the L2 in a 64-B atomic request (TCC_EA0_ATOMIC %.3g per launch, %s: %.2f doubles each; "
                                 "6.41 is what whole rows of the sphere give) against %.3g per second the memory side takes (%s; "
                                 "`atomic_requests.frac`): worth the 9 %% the ablation shows, not more.  Third, the four divergent 16-B rate-table "
                                 "gathers per cell: +20 %% on a field with fronts (`evolving_state.roofline_frac`).  HBM bytes are not close "
                                 "(`traffic` = 0.97 x algorithmic).  DESIGN.md 7"
                                 % (rt_counters.get("SQ_INSTS_VALU", float("nan")), rt_counters.get("TCC_EA0_ATOMIC_sum", float("nan")), PMC_SUMMARY,
                                    gamma_cells / rt_counters.get("TCC_EA0_ATOMIC_sum", float("nan")), ATOMIC_REQUEST_CEILING,
                                    ATOMIC_CEILING_SOURCE))
                                if (default_job and "TCC_EA0_ATOMIC_sum" in rt_counters) else None,
        },
        "roofline_kernels": [
            {"kernel": "raytrace_octant_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "bytes_per_unit": RT_BYTES_PER_UPDATE,
             "units_per_launch": gamma_cells, "avg_launch_ms": rt_ms / max(rt_n, 1), "share_of_step": (rt_ms / n_timed) / (elapsed / K * 1e3),
             "counter_bytes": pmc_traffic_bytes("raytrace_octant_kernel", pmc_summary),
             # secondary line ("FP64-vector utilisation ... may make the raytrace compute-bound before HBM"): VALU
             # wave-instructions per launch from the committed counters over this run's launch time, against one wave-instruction
             # per SIMD every 4 cycles (1024 SIMDs at the 2.4 GHz maximum clock, MI355X_MICROARCH.md)
             "valu_issue": ({"wave_instructions_per_launch": rt_counters["SQ_INSTS_VALU"],
                             "achieved_per_s": rt_counters["SQ_INSTS_VALU"] / rt_launch_s, "peak_per_s": 1024 * 2.4e9 / 4.0,
                             "frac": rt_counters["SQ_INSTS_VALU"] / rt_launch_s / (1024 * 2.4e9 / 4.0), "source": PMC_SUMMARY}
                            if (default_job and "SQ_INSTS_VALU" in rt_counters and rt_n) else None),
             # one of the launch's three co-limits (`roofline.binding_resource`): 64-B atomic requests leaving the L2 (committed counters)
             # over this run's launch time, against the rate a bare stream of such atomics reaches (tools/micro/atomic_rate.hip)
             "atomic_requests": ({"requests_per_launch": rt_counters["TCC_EA0_ATOMIC_sum"],
                                  "achieved_per_s": rt_counters["TCC_EA0_ATOMIC_sum"] / rt_launch_s, "ceiling_per_s": ATOMIC_REQUEST_CEILING,
                                  "frac": rt_counters["TCC_EA0_ATOMIC_sum"] / rt_launch_s / ATOMIC_REQUEST_CEILING,
                                  "source": PMC_SUMMARY + ", " + ATOMIC_CEILING_SOURCE}
                                 if (default_job and "TCC_EA0_ATOMIC_sum" in rt_counters and rt_n) else None)},
            {"kernel": "chemistry_tile_kernel", "bound": "hbm", "achieved": ch_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (ch_achieved / HBM_PEAK_GBS) if ch_achieved else None, "bytes_per_unit": CHEM_BYTES_PER_UPDATE,
             "units_per_launch": chem_cells, "avg_launch_ms": ch_ms / max(ch_n, 1), "share_of_step": (ch_ms / n_timed) / (elapsed / K * 1e3),
             "counter_bytes": pmc_traffic_bytes("chemistry_tile_kernel", pmc_summary),
             "bytes_moved_per_unit": CHEM_FUSED_BYTES_PER_UPDATE if fused_pass else CHEM_BYTES_PER_UPDATE,
             "moved_GBs": ch_actual,
             "note": "the fused pass also folds the two rate accumulators, writes nHI in both layouts for the next raytrace and "
                     "zeroes the accumulators: 88 B per cell move through HBM (uniform temperature: that grid is not read; the folded rates are not stored), of which 56 B are "
                     "the chemistry's own algorithmic traffic"},
        ],
        "kernels_ms_per_step": {
            "raytrace": rt_ms / n_timed, "chemistry": ch_ms / n_timed, "prepare_nhi": pr_ms / n_timed, "fold_phi_t": fi_ms / n_timed,
            "chemistry_achieved_GBs": ch_achieved,
        },
        "raytrace_ns_per_source_per_insphere_cell": (rt_ms / max(rt_n, 1)) * 1e6 / (max(n_local, 1) * insphere),
    }

    if comm is not None:
        ms_step = elapsed / K * 1e3
        out["one_gpu_same_workload_ms_per_step"] = one_gpu_ms
        out["speedup_vs_one_gpu"] = (one_gpu_ms / ms_step) if one_gpu_ms else None
        out["one_gpu_same_workload_note"] = ("rank 0 alone, the whole source list of THIS workload through the one-GPU loop "
                                             "(asora_evolve_enqueue + one poll per %d steps): %s, while the other "
                                             "ranks waited at a barrier; speedup = that / ms_per_step" % (poll_every, one_gpu_how))
        out["phases_ms"] = phases_ms
        # the host's share: what it takes to ISSUE one iteration of the sharded loop (library calls + torch.distributed calls; the
        # device runs behind, a batch of iterations per poll) -- must stay below the iteration's GPU time or the GPU starves
        out["host_issue_ms_per_step"] = (state["host_enqueue_s"] / state["host_enqueued"] * 1e3) if state["host_enqueued"] else None
        out["phases_note"] = ("mean per step over the timed regions, MAX over the ranks; slab exchange: spans between HIP events on the "
                              "library's stream (wait_rates_add = what the rate exchange left un-hidden + the adds; xh_av_exchange_nhi is "
                              "serial by construction), all-reduce path: wall clock between the host synchronisations of its three calls")
        out["measured_link_GBs"] = links
        out["configs4"] = configs4
    if world == 1 and (args.evolving_state == 1 or (args.evolving_state < 0 and default_job)):
        try:
            out["evolving_state"] = evolving_state(lib, p, _capi, N, args.R, thin, thick, dlog, numtau)
        except Exception as e:   # reporting only
            out["evolving_state"] = {"failed": f"{type(e).__name__}: {e}"}
    if world == 1 and args.cpu_sources > 0:
        try:
            use_ref, ns, t_rt, t_chem = cpu_baseline(args.workload, N, ndens, xh, temp, dr, pos, flux, thin, thick,
                                                     dlog, args.R, args.nsrc, args.cpu_sources)
            per_src_gamma = gamma_cells / args.nsrc
            t_job = t_rt * (args.nsrc / ns) + t_chem
            out["cpu_baseline"] = {
                "value": (per_src_gamma * args.nsrc + N ** 3) / t_job,
                "unit": "cell-updates/s",
                "cores": 1,
                "kind": "reference" if use_ref else "port",
                "sample": (f"{ns} of {args.nsrc} sources raytraced ({t_rt:.2f} s, cube +-{int(np.ceil(args.R))} per source, "
                           f"single-threaded as the reference is) + one global_pass over {N}^3 ({t_chem:.2f} s); "
                           f"raytrace time scaled x{args.nsrc / ns:.1f} to the job"),
                "raytrace_s_per_source": t_rt / ns,
                "chemistry_s_per_pass": t_chem,
            }
        except Exception as e:   # the baseline is reporting only; never let it hide the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
        if isinstance(out.get("cpu_baseline"), dict):
            out["cpu_baseline"].update({"host_cores_total": topo["logical_cpus"], "sockets": topo["sockets"],
                                        "physical_cores_per_socket": topo["physical_cores_per_socket"],
                                        "cpus_this_process_may_use": topo["usable_cpus"]})
        if cpu_workers:
            try:
                wall, res = run_cpu_workers(cpu_workers)
                ns_all = sum(r["sources"] for r in res)
                wall_rt = max(r["t_rt"] for r in res)
                wall_chem = max(r["t_chem"] for r in res)
                t_job = wall_rt * (args.nsrc / ns_all) + wall_chem
                socket_cores = topo["physical_cores_per_socket"]
                whole = cpu_cores >= socket_cores
                v = (gamma_cells + N ** 3) / t_job
                out["cpu_baseline"]["all_cores"] = {
                    "value": v, "unit": "cell-updates/s", "cores": cpu_cores,
                    "kind": "reference" if all(r["reference"] for r in res) else "port",
                    "socket_cores": socket_cores, "covers_one_socket": bool(whole),
                    # fewer workers than the socket has cores (CPU share or memory of this process): the socket figure is the
                    # measured one scaled by the core ratio -- an upper bound, memory bandwidth does not scale that well
                    "one_socket_value": v if whole else v * socket_cores / cpu_cores,
                    "one_socket_value_is": "measured" if whole else f"extrapolated x{socket_cores / cpu_cores:.2f} from {cpu_cores} cores",
                    "gpu_over_one_socket": value / (v if whole else v * socket_cores / cpu_cores),
                    "sample": (f"{cpu_cores} independent single-threaded processes (the reference is single-threaded, sources are "
                               f"independent), one per physical core of ONE socket ({socket_cores} cores; host: {topo['sockets']} socket(s), "
                               f"{topo['logical_cpus']} logical CPUs, this process may use {topo['usable_cpus']}"
                               + (f"; {cpu_note}" if cpu_note else "") + f"): {ns_all} of {args.nsrc} "
                               f"sources raytraced ({wall_rt:.2f} s for the slowest worker) and one global_pass over {N}^3 split "
                               f"into {cpu_cores} slabs ({wall_chem:.2f} s); raytrace time scaled x{args.nsrc / ns_all:.1f}; inputs "
                               "shared read-only between the workers"),
                }
            except Exception as e:
                out["cpu_baseline"]["all_cores"] = {"value": None, "sample": f"failed: {type(e).__name__}: {e}"}
    if cpu_workers:                       # never leave workers or their shared-memory files behind
        for pr in cpu_workers:
            if pr.poll() is None:
                pr.kill()
    if cpu_dir:
        import shutil
        shutil.rmtree(cpu_dir, ignore_errors=True)
    p.device_close()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    print(json.dumps(out), flush=True)
    if comm is not None:
        os.dup2(2, 1)              # nothing after the JSON line may reach stdout
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
