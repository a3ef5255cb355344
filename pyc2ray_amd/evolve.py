"""One C2-Ray time step: raytrace all sources, solve the chemistry, iterate to convergence.

Same two entry points, argument lists, return values, log lines and convergence logic as the
reference (pyc2ray/evolve.py:38-245 ``evolve3D``, :249-498 ``evolve3D_MPI``).  What differs is where
the data lives: the reference re-uploads xh_av, downloads phi_ion, runs the chemistry on one CPU core
and transposes xh_av back on every iteration (evolve.py:187,200,210,240); here ndens, temp, xh, xh_av,
xh_intermed and phi_ion stay on the MI355X for the whole step, the chemistry is a HIP kernel, and
three scalars (conv_flag, sum x, sum 1-x) cross PCIe per iteration.
"""
import array
import os
import time

import numpy as np

from . import _capi, _residency
from .asora_core import cuda_is_init
from .load_extensions import load_asora, load_c2ray
from .utils import printlog
from .utils.logutils import printlog_lines
from .utils.sourceutils import format_sources

__all__ = ['evolve3D', 'evolve3D_MPI', 'evolve3D_resident']

#: outer iterations enqueued per host round trip of the single-GPU loop (the device evaluates the convergence test
#: itself; launches enqueued beyond convergence do nothing)
EVOLVE_BATCH = int(os.environ.get("PYC2RAY_AMD_EVOLVE_BATCH", "8"))


def _has_transposed_twins(libasora):
    """The sharded / all-reduce device loops need the [k][j][i] twins of the grids (asora_evolve_begin_slab fails without:
    ASORA_OPT_Z_TRANSPOSED = 0 is a diagnostic setting)."""
    get = getattr(libasora, "get_option", None)
    return True if get is None else get(_capi.OPT_Z_TRANSPOSED) != 0


def _comm_backend(comm):
    """"nccl" / "gloo" of a pyc2ray_amd.dist.TorchComm-shaped communicator (public `backend`; older objects: `_backend()`);
    anything else exchanges through the host like gloo."""
    b = getattr(comm, "backend", None)
    if callable(b):
        b = b()
    if b is None and hasattr(comm, "_backend"):
        b = comm._backend()
    return b if isinstance(b, str) else "gloo"


def _next_batch(history, batch_max, conv_criterion, convergence_fraction):
    """How many iterations of a multi-rank device loop to enqueue before the next poll.  The launches of an iteration are gated
    by the device's `done` flag, its collectives are not: an iteration enqueued beyond convergence still exchanges its planes
    (slab path) or all-reduces the N^3 out-box (all-reduce path).  So the batch follows the distance to the test of
    evolve.py:232: `history` = the rows polled so far (conv_flag, sum1, sum0, rel1, rel0); both criteria decay roughly
    geometrically from iteration to iteration, and the batch is the number of iterations the faster of the two still needs at
    the rate of the last two rows -- at least 1, at most batch_max, and 2 while there is nothing to extrapolate from.  Every
    rank sees the same rows, hence the same batch."""
    import math
    if batch_max <= 1:
        return 1
    if len(history) < 2:
        return min(2, batch_max)

    def remaining(prev, last, target):
        if last < target:
            return 0
        if not (0.0 < last < prev):
            return batch_max
        rate = last / prev
        return max(1, int(math.ceil(math.log(max(target, 1e-300) / last) / math.log(rate))))

    (f0, _, _, r10, r00), (f1, _, _, r11, r01) = history[-2], history[-1]
    by_change = max(remaining(r10, r11, convergence_fraction), remaining(r00, r01, convergence_fraction))
    by_count = remaining(float(f0), float(f1), float(conv_criterion)) if conv_criterion > 0 else batch_max
    return max(1, min(batch_max, by_change, by_count))


def _agree_on_convergence(comm, converged):
    """Rank 0 decides, everyone follows (pyc2ray/evolve.py:484-489).  Every rank derives `converged` from the same
    summed rates, but a collective that sums in a rank-dependent order could leave them one ulp apart; a rank that
    left the loop alone would hang the others in the next collective."""
    flag = array.array('i', [int(bool(converged))])
    comm.Bcast(flag, root=0)
    return bool(flag[0])


def _evolve_cpu_semantics(dt, dr, src_flux, src_pos, max_subbox, subboxsize, loss_fraction, temp, ndens, xh,
                          photo_thin_table, photo_thick_table, minlogtau, dlogtau, R_max_LLS, convergence_fraction,
                          sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet,
                          use_mpi=None, comm=None, rank=0, nprocs=1):
    """The use_gpu=False branch of the reference (pyc2ray/evolve.py:168-245, :401-498): per iteration one pass of
    the CPU library's raytracer -- cubic sub-boxes grown until the photon loss is below loss_fraction, photon-loss
    statistics, Fortran-flavoured constants, every source rated with the flux of the last one as the Fortran does
    -- and one global_pass.  Both are evaluated on the GPU, and like the use_gpu=True loop this one keeps the grids
    on the device for the whole step (the reference's host round trips are what
    ``libc2ray.raytracing.do_all_sources`` / ``libc2ray.chemistry.global_pass`` of this package still offer)."""
    _residency.reclaim()              # this step overwrites device grids a resident C2Ray object may be relying on
    libasora = load_asora()
    distributed = bool(use_mpi) and comm is not None and nprocs > 1
    NumSrc = src_flux.shape[0]
    N = temp.shape[0]
    NumCells = N * N * N
    NumTau = photo_thin_table.shape[0]
    conv_criterion = min(int(convergence_fraction * NumCells), (NumSrc - 1) / 3)        # evolve.py:127
    prev_sum_xh1_int = 2 * NumCells
    prev_sum_xh0_int = 2 * NumCells
    converged = False
    niter = 0
    if distributed:                                                                     # evolve.py:360-371
        perrank = NumSrc // nprocs
        i_start = int(rank * perrank)
        i_end = int((rank + 1) * perrank) if rank != nprocs - 1 else NumSrc
        my_flux, my_pos = src_flux[i_start:i_end], np.asarray(src_pos)[:, i_start:i_end]
        printlog(f"...rank={rank:n} has {i_end - i_start:n} sources.", logfile, quiet)
    else:
        my_flux, my_pos = src_flux, np.asarray(src_pos)
    n_local = my_flux.shape[0]

    # this branch has no device_init of its own in the reference: the library sets itself up for the mesh
    libasora.device_init_auto(N)
    libasora.photo_table_to_device(photo_thin_table, photo_thick_table, NumTau)
    srcpos_flat, normflux_flat = format_sources(my_pos, my_flux)
    libasora.source_data_to_device(srcpos_flat, normflux_flat, n_local)
    libasora.grid_to_device(_capi.GRID_NDENS, ndens)
    libasora.grid_to_device(_capi.GRID_TEMP, temp)
    libasora.grid_to_device(_capi.GRID_XH, xh)
    libasora.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)          # xh_av = copy(xh)        evolve.py:136
    libasora.grid_copy(_capi.GRID_XH_INTERMED, _capi.GRID_XH)    # xh_intermed = copy(xh)  evolve.py:137
    if rank == 0:
        printlog("Calling evolve3D...", logfile, quiet)
        printlog(f"dr [Mpc]: {dr/3.086e24:.3e}", logfile, quiet)
        printlog(f"dt [years]: {dt/3.15576E+07:.3e}", logfile, quiet)
        printlog(f"Running on {NumSrc:n} source(s), total normalized ionizing flux: {src_flux.sum():.2e}", logfile, quiet)
        mean_ndens = libasora.grid_sum(_capi.GRID_NDENS) / NumCells
        mean_xh = libasora.grid_sum(_capi.GRID_XH) / NumCells
        printlog(f"Mean density (cgs): {mean_ndens:.3e}, Mean ionized fraction: {mean_xh:.3e}", logfile, quiet)
        printlog(f"Convergence Criterion (Number of points): {conv_criterion : n}", logfile, quiet, end='\n\n')
    while not converged:
        niter += 1
        trt0 = time.time()
        printlog("Doing Raytracing...", logfile, quiet, ' ')
        nsubbox, photonloss = libasora.subbox_raytrace_device(max_subbox, subboxsize, loss_fraction, R_max_LLS, sig, dr,
                                                              minlogtau, dlogtau, NumTau, 0, n_local)
        printlog(f"took {(time.time()-trt0) : .1f} s.", logfile, quiet)
        printlog(f"Average number of subboxes: {nsubbox/max(n_local, 1):n}, Total photon loss: {photonloss:.3e}",
                 logfile, quiet)
        if distributed:                                                                 # evolve.py:433-437
            _allreduce_phi(libasora, N, use_mpi, comm, rank)
        tch0 = time.time()
        if rank == 0:
            printlog("Doing Chemistry...", logfile, quiet, ' ')
        conv_flag, sum_xh1_int, sum_xh0_int = libasora.chemistry_device(dt, bh00, albpow, colh0, temph0, abu_c)
        if rank == 0:
            printlog(f"took {(time.time()-tch0) : .1f} s.", logfile, quiet)
        rel_change_xh1 = np.abs((sum_xh1_int - prev_sum_xh1_int) / sum_xh1_int) if sum_xh1_int > 0.0 else 1.0
        rel_change_xh0 = np.abs((sum_xh0_int - prev_sum_xh0_int) / sum_xh0_int) if sum_xh0_int > 0.0 else 1.0
        if rank == 0:
            printlog(f"Number of non-converged points: {conv_flag} of {NumCells} ({conv_flag / NumCells * 100 : .3f} % ), "
                     f"Relative change in ionfrac: {rel_change_xh1 : .2e}", logfile, quiet)
        converged = (conv_flag < conv_criterion) or ((rel_change_xh1 < convergence_fraction) and
                                                     (rel_change_xh0 < convergence_fraction))
        if distributed:
            converged = _agree_on_convergence(comm, converged)
        prev_sum_xh1_int = sum_xh1_int
        prev_sum_xh0_int = sum_xh0_int
    if rank == 0:
        printlog("Multiple source convergence reached.", logfile, quiet)
    # Fortran-ordered results, as the reference's CPU branch returns them (evolve.py:178)
    xh_new = libasora.grid_to_host(_capi.GRID_XH_INTERMED, libasora.host_empty((N, N, N), order='F'))
    phi_ion = libasora.grid_to_host(_capi.GRID_PHI_ION, libasora.host_empty((N, N, N), order='F'))
    _evolve.last_niter = niter
    return xh_new, phi_ion


def _allreduce_phi(libasora, N, use_mpi, comm, rank):
    """Sum the per-rank photo-ionisation rate grids (evolve.py:433-437: Reduce to root + Bcast)."""
    if hasattr(comm, "allreduce_device_grid"):
        # pyc2ray_amd.dist.TorchComm: RCCL all-reduce on the device-resident grid (or gloo via host)
        comm.allreduce_device_grid(libasora, _capi.GRID_PHI_ION, N)
        return
    # mpi4py communicator, host-staged like the reference
    phi = libasora.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
    if rank == 0:
        comm.Reduce(use_mpi.IN_PLACE, [phi, use_mpi.DOUBLE], op=use_mpi.SUM, root=0)
    else:
        comm.Reduce([phi, use_mpi.DOUBLE], None, op=use_mpi.SUM, root=0)
    comm.Bcast([phi, use_mpi.DOUBLE], root=0)
    libasora.grid_to_device(_capi.GRID_PHI_ION, phi)


def _device_loop(libasora, chem, R_max_LLS, sig, dr, minlogtau, dlogtau, NumTau, NumSrc_local, conv_criterion,
                 convergence_fraction, NumCells, logfile, quiet):
    """One GPU: the whole loop of a time step on the device (include/asora_hip.h, asora_evolve_*).  An iteration is
    the raytrace plus ONE pass over the grids (rates folded, chemistry, nHI for the next trace, accumulators zeroed); the
    convergence test of evolve.py:216-236 is evaluated on the device, so EVOLVE_BATCH iterations are enqueued per host
    round trip and those beyond convergence do nothing.  NDENS, TEMP, XH and the sources must be on the device.
    Returns (outer iterations, sum of xh_intermed of the last iteration)."""
    libasora.evolve_begin(*chem, R_max_LLS, sig, dr, minlogtau, dlogtau, NumTau, 0, NumSrc_local,
                          conv_criterion, convergence_fraction)
    batch = max(1, min(EVOLVE_BATCH, 32))
    niter, converged, sum_xh1 = 0, False, 0.0
    while not converged:
        t0 = time.time()
        libasora.evolve_enqueue(batch)
        _, converged, rows = libasora.evolve_poll(batch)
        per_iteration = (time.time() - t0) / max(len(rows), 1)
        lines = []
        for conv_flag, sum_xh1, _s0, rel_change_xh1, _rel0 in rows:
            niter += 1
            conv_flag = int(conv_flag)
            lines += [("Doing Raytracing...", ' '), (f"took {per_iteration : .1f} s.", '\n'),
                      ("Doing Chemistry...", ' '),
                      ("took  0.0 s. (fused with the raytrace on the device: the time above is for both)", '\n'),
                      (f"Number of non-converged points: {conv_flag} of {NumCells} ({conv_flag / NumCells * 100 : .3f} % ), "
                       f"Relative change in ionfrac: {rel_change_xh1 : .2e}", '\n')]
        printlog_lines(lines, logfile, quiet)
    return niter, float(sum_xh1)


def evolve3D_resident(dt, dr, src_flux, src_pos, uploads, N, photo_thin_table, minlogtau, dlogtau, R_max_LLS,
                      convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile="pyC2Ray.log", quiet=False):
    """evolve3D for a caller that keeps the grids on the device between time steps (the C2Ray class with
    ``device_resident = True``): same loop, log lines and results as :func:`evolve3D` with ``use_gpu=True``, but only the
    grids in ``uploads`` ({grid selector: host array}, those the caller changed on the host) cross PCIe, and nothing
    comes back: afterwards XH_INTERMED == XH == the new ionised fraction and PHI_ION the rates, on the device
    (``libasora.grid_to_host`` fetches them when someone looks).  Returns the number of outer iterations."""
    if not cuda_is_init():
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
    libasora = load_asora()
    NumSrc = src_flux.shape[0]
    NumCells = N * N * N
    NumTau = photo_thin_table.shape[0]
    conv_criterion = min(int(convergence_fraction * NumCells), (NumSrc - 1) / 3)          # evolve.py:127
    srcpos_flat, normflux_flat = format_sources(np.asarray(src_pos), src_flux)
    libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc)
    for which, grid in uploads.items():
        libasora.grid_to_device(which, grid)
    printlog("Copied source data to device.", logfile, quiet)
    printlog("Calling evolve3D...", logfile, quiet)
    printlog(f"dr [Mpc]: {dr/3.086e24:.3e}", logfile, quiet)
    printlog(f"dt [years]: {dt/3.15576E+07:.3e}", logfile, quiet)
    printlog(f"Running on {NumSrc:n} source(s), total normalized ionizing flux: {src_flux.sum():.2e}", logfile, quiet)
    mean_ndens = libasora.grid_sum(_capi.GRID_NDENS) / NumCells
    mean_xh = libasora.grid_sum(_capi.GRID_XH) / NumCells
    printlog(f"Mean density (cgs): {mean_ndens:.3e}, Mean ionized fraction: {mean_xh:.3e}", logfile, quiet)
    printlog(f"Convergence Criterion (Number of points): {conv_criterion : n}", logfile, quiet, end='\n\n')
    niter, _ = _device_loop(libasora, (dt, bh00, albpow, colh0, temph0, abu_c), R_max_LLS, sig, dr, minlogtau, dlogtau,
                            NumTau, NumSrc, conv_criterion, convergence_fraction, NumCells, logfile, quiet)
    printlog("Multiple source convergence reached.", logfile, quiet)
    libasora.grid_copy(_capi.GRID_XH, _capi.GRID_XH_INTERMED)       # the next step starts from the new ionised fraction
    _evolve.last_niter = niter
    return niter


def _evolve(dt, dr, src_flux, src_pos, use_gpu, temp, ndens, xh, photo_thin_table, minlogtau, dlogtau,
            R_max_LLS, convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet,
            use_mpi=None, comm=None, rank=0, nprocs=1):
    if use_gpu and not cuda_is_init():
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
    _residency.reclaim()              # this step overwrites device grids a resident C2Ray object may be relying on
    distributed = bool(use_mpi) and comm is not None and nprocs > 1
    libasora = load_asora()

    NumSrc = src_flux.shape[0]          # number of sources
    N = temp.shape[0]                   # mesh size
    NumCells = N * N * N
    NumTau = photo_thin_table.shape[0]  # evolve.py:124 (the table LENGTH is what the reference passes)

    # Convergence criterion, evolve.py:127 (computed from the TOTAL source count, evolve.py:346)
    conv_criterion = min(int(convergence_fraction * NumCells), (NumSrc - 1) / 3)

    prev_sum_xh1_int = 2 * NumCells
    prev_sum_xh0_int = 2 * NumCells
    converged = False
    niter = 0

    # source shard of this rank, evolve.py:360-371
    # (the sharded device loop needs the [k][j][i] twins, asora_evolve_begin_slab: not with ASORA_OPT_Z_TRANSPOSED = 0)
    slab = (distributed and hasattr(comm, "slab_enqueue") and getattr(comm, "exchange", "") == "slab"
            and not getattr(comm, "overlap", False) and _has_transposed_twins(libasora))
    plan = None
    all_pos, all_flux = np.asarray(src_pos), src_flux
    if slab:
        # the same contiguous blocks, of the list ordered by the first coordinate: a rank's rates then live on the
        # planes within R of its slab of sources, and only those planes are exchanged (pyc2ray_amd.dist.SlabPlan)
        from .dist import SlabPlan
        all_pos, all_flux, bounds = comm.shard_sources_by_slab(all_pos, all_flux, nprocs)
        i_start, i_end = bounds[rank], bounds[rank + 1]
        plan = SlabPlan(N, nprocs, R_max_LLS, [all_pos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(nprocs)])
    elif distributed:
        perrank = NumSrc // nprocs
        i_start = int(rank * perrank)
        i_end = int((rank + 1) * perrank) if rank != nprocs - 1 else NumSrc
    else:
        i_start, i_end = 0, NumSrc
    NumSrc_local = i_end - i_start
    my_pos, my_flux = all_pos[:, i_start:i_end], all_flux[i_start:i_end]
    # pipelined raytrace + all-reduce (pyc2ray_amd.dist, opt-in): the shard is traced in order of the first coordinate
    pipelined = distributed and getattr(comm, "overlap", False) and hasattr(comm, "raytrace_and_allreduce")
    src_i0 = None
    if pipelined:
        my_pos, my_flux = comm.sort_sources_for_overlap(my_pos, my_flux)
        src_i0 = np.asarray(my_pos[0]).astype(np.int64) - 1
    srcpos_flat, normflux_flat = format_sources(my_pos, my_flux)
    if distributed:
        printlog(f"...rank={rank:n} has {NumSrc_local:n} sources.", logfile, quiet)

    # Everything the step needs goes to the device once (evolve.py:136-155 keeps host copies instead)
    libasora.source_data_to_device(srcpos_flat, normflux_flat, NumSrc_local)
    libasora.grid_to_device(_capi.GRID_NDENS, ndens)
    libasora.grid_to_device(_capi.GRID_TEMP, temp)
    libasora.grid_to_device(_capi.GRID_XH, xh)
    if distributed:
        libasora.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)          # xh_av = copy(xh)        evolve.py:136
        libasora.grid_copy(_capi.GRID_XH_INTERMED, _capi.GRID_XH)    # xh_intermed = copy(xh)  evolve.py:137
    else:
        printlog("Copied source data to device.", logfile, quiet)

    if rank == 0:
        if distributed:
            printlog(f"Calling evolve3D with {nprocs:n} MPI-processors...", logfile, quiet)
        else:
            printlog("Calling evolve3D...", logfile, quiet)
        printlog(f"dr [Mpc]: {dr/3.086e24:.3e}", logfile, quiet)
        printlog(f"dt [years]: {dt/3.15576E+07:.3e}", logfile, quiet)
        printlog(f"Running on {NumSrc:n} source(s), total normalized ionizing flux: {src_flux.sum():.2e}", logfile, quiet)
        # the two means of evolve.py:160, summed on the device from the grids just uploaded
        mean_ndens = libasora.grid_sum(_capi.GRID_NDENS) / NumCells
        mean_xh = libasora.grid_sum(_capi.GRID_XH) / NumCells
        printlog(f"Mean density (cgs): {mean_ndens:.3e}, Mean ionized fraction: {mean_xh:.3e}", logfile, quiet)
        printlog(f"Convergence Criterion (Number of points): {conv_criterion : n}", logfile, quiet, end='\n\n')

    chem = (dt, bh00, albpow, colh0, temph0, abu_c)
    if not distributed:
        niter, _ = _device_loop(libasora, chem, R_max_LLS, sig, dr, minlogtau, dlogtau, NumTau, NumSrc_local, conv_criterion,
                                convergence_fraction, NumCells, logfile, quiet)
        converged = True

    if slab:
        # the device-resident loop, sharded (pyc2ray_amd.dist.TorchComm.slab_*): per iteration the trace, the rates to the
        # owners of the planes, ONE fused pass on the own slab, xh_av back, and the convergence test on the device behind the
        # in-place all-reduce of its three sums -- identical bits, hence the same decision, on every rank.  With RCCL a batch of
        # iterations is enqueued per host round trip (launches beyond convergence do nothing); with gloo every exchange
        # goes through the host anyway and the batch is one.
        comm.slab_begin(libasora, plan, N, R_max_LLS, sig, dr, NumSrc_local, minlogtau, dlogtau, NumTau, chem,
                        conv_criterion, convergence_fraction)
        batch_max = max(1, min(EVOLVE_BATCH, 32)) if _comm_backend(comm) == "nccl" else 1
        history = []
        while not converged:
            trt0 = time.time()
            batch = _next_batch(history, batch_max, conv_criterion, convergence_fraction)
            comm.slab_enqueue(libasora, batch)
            _, converged, rows = comm.slab_poll(libasora, batch)
            history += list(rows)
            per_iteration = (time.time() - trt0) / max(len(rows), 1)
            lines = []
            for conv_flag, _s1, _s0, rel_change_xh1, _rel0 in rows:
                niter += 1
                conv_flag = int(conv_flag)
                lines += [(f"Doing Raytracing and Chemistry, slab-wise (rank={rank:n})...", ' '),
                          (f"rank={rank:n} took {per_iteration : .1e} s.", '\n')]
                if rank == 0:
                    lines += [(f"Number of non-converged points: {conv_flag} of {NumCells} ({conv_flag / NumCells * 100 : .3f} % ), "
                               f"Relative change in ionfrac: {rel_change_xh1 : .2e}", '\n')]
            printlog_lines(lines, logfile, quiet)

    # full-grid all-reduce on the same device-resident loop (TorchComm.reduce_begin): trace, fold, all-reduce of the rate grid in
    # place, ONE fused pass on the whole grid on every rank, test on the device -- batches of iterations per host round trip
    # (asora_evolve_begin_slab needs the [k][j][i] twins: with ASORA_OPT_Z_TRANSPOSED = 0 the three-call loop below runs)
    reduce_loop = (distributed and not slab and not pipelined and getattr(comm, "device_loop", False)
                   and hasattr(comm, "reduce_begin") and hasattr(libasora, "evolve_slab_fold_all")
                   and _has_transposed_twins(libasora))
    if reduce_loop:
        comm.reduce_begin(libasora, N, R_max_LLS, sig, dr, NumSrc_local, minlogtau, dlogtau, NumTau, chem, conv_criterion,
                          convergence_fraction)
        batch_max = max(1, min(EVOLVE_BATCH, 32)) if _comm_backend(comm) == "nccl" else 1
        history = []
        while not converged:
            trt0 = time.time()
            # (an all-reduce is not gated by the device's `done` flag: every iteration enqueued beyond convergence still sums
            #  the N^3 out-box over the ranks, so the batch shrinks as the test comes within reach)
            batch = _next_batch(history, batch_max, conv_criterion, convergence_fraction)
            comm.slab_enqueue(libasora, batch)
            _, converged, rows = comm.slab_poll(libasora, batch)
            history += list(rows)
            per_iteration = (time.time() - trt0) / max(len(rows), 1)
            lines = []
            for conv_flag, _s1, _s0, rel_change_xh1, _rel0 in rows:
                niter += 1
                conv_flag = int(conv_flag)
                lines += [(f"Doing Raytracing, all-reduce and Chemistry (rank={rank:n})...", ' '),
                          (f"rank={rank:n} took {per_iteration : .1e} s.", '\n')]
                if rank == 0:
                    lines += [(f"Number of non-converged points: {conv_flag} of {NumCells} ({conv_flag / NumCells * 100 : .3f} % ), "
                               f"Relative change in ionfrac: {rel_change_xh1 : .2e}", '\n')]
            printlog_lines(lines, logfile, quiet)

    while distributed and not slab and not reduce_loop and not converged:
        niter += 1

        # (1) raytracing, evolve.py:174-196
        trt0 = time.time()
        if pipelined:
            # raytrace, sum over ranks and chemistry slab by slab (pyc2ray_amd.dist): steps (1) and (2) in one
            printlog(f"Doing Raytracing and Chemistry, pipelined (rank={rank:n})...", logfile, quiet, ' ')
            conv_flag, sum_xh1_int, sum_xh0_int = comm.raytrace_and_allreduce(
                libasora, N, R_max_LLS, sig, dr, NumSrc_local, minlogtau, dlogtau, NumTau, src_i0=src_i0, chemistry=chem)
            printlog(f"rank={rank:n} took {(time.time()-trt0) : .1e} s.", logfile, quiet)
        else:
            printlog(f"Doing Raytracing (rank={rank:n})...", logfile, quiet, ' ')
            libasora.raytrace_device(R_max_LLS, sig, dr, 0, NumSrc_local, minlogtau, dlogtau, NumTau)
            libasora.synchronize()
            printlog(f"rank={rank:n} took {(time.time()-trt0) : .1e} s.", logfile, quiet)
            _allreduce_phi(libasora, N, use_mpi, comm, rank)

            # (2) chemistry, evolve.py:207-211.  Every rank runs it on the identical summed rates
            # (the reference runs it on rank 0 and broadcasts two N^3 grids, evolve.py:439-481).
            tch0 = time.time()
            if rank == 0:
                printlog("Doing Chemistry...", logfile, quiet, ' ')
            conv_flag, sum_xh1_int, sum_xh0_int = libasora.chemistry_device(*chem)
            if rank == 0:
                printlog(f"took {(time.time()-tch0) : .1f} s.", logfile, quiet)

        # (3) global convergence, evolve.py:216-236
        if sum_xh1_int > 0.0:
            rel_change_xh1 = np.abs((sum_xh1_int - prev_sum_xh1_int) / sum_xh1_int)
        else:
            rel_change_xh1 = 1.0
        if sum_xh0_int > 0.0:
            rel_change_xh0 = np.abs((sum_xh0_int - prev_sum_xh0_int) / sum_xh0_int)
        else:
            rel_change_xh0 = 1.0

        if rank == 0:
            printlog(f"Number of non-converged points: {conv_flag} of {NumCells} ({conv_flag / NumCells * 100 : .3f} % ), "
                     f"Relative change in ionfrac: {rel_change_xh1 : .2e}", logfile, quiet)

        converged = (conv_flag < conv_criterion) or ((rel_change_xh1 < convergence_fraction) and
                                                     (rel_change_xh0 < convergence_fraction))
        converged = _agree_on_convergence(comm, converged)            # evolve.py:484-489
        prev_sum_xh1_int = sum_xh1_int
        prev_sum_xh0_int = sum_xh0_int

    if rank == 0:
        printlog("Multiple source convergence reached.", logfile, quiet)
    if slab:       # every rank returns the whole fields (evolve.py:480-481,497): collect the owners' slabs
        comm.slab_gather(libasora, plan, _capi.GRID_XH_INTERMED, N)
        comm.slab_gather(libasora, plan, _capi.GRID_PHI_ION, N)
    # laid out like `xh`, as np.empty_like would; in page-locked memory (pyc2ray_amd/_pinned.py)
    like_xh = 'F' if (xh.flags.f_contiguous and not xh.flags.c_contiguous) else 'C'
    xh_new = libasora.grid_to_host(_capi.GRID_XH_INTERMED, libasora.host_empty((N, N, N), order=like_xh))
    phi_ion = libasora.grid_to_host(_capi.GRID_PHI_ION, libasora.host_empty((N, N, N)))
    _evolve.last_niter = niter
    return xh_new, phi_ion


def evolve3D(dt, dr,
             src_flux, src_pos,
             use_gpu, max_subbox, subboxsize, loss_fraction,
             temp, ndens, xh,
             photo_thin_table, photo_thick_table,
             minlogtau, dlogtau,
             R_max_LLS, convergence_fraction,
             sig, bh00, albpow, colh0, temph0, abu_c,
             logfile="pyC2Ray.log", quiet=False):
    """Evolve the ionised fraction of the whole grid over one time step.

    Parameters have the reference's meaning (pyc2ray/evolve.py:49-109): dt [s], dr [cm],
    src_flux (numsrc) in units of 1e48 s^-1, src_pos (3,numsrc) 1-based, temp/ndens/xh (N,N,N),
    tables as copied to the GPU beforehand with photo_table_to_device(), R_max_LLS in cells.
    max_subbox, subboxsize and loss_fraction only concern the reference's CPU raytracer and have no
    effect with use_gpu=True.  use_gpu=False selects that raytracer's semantics (cubic sub-boxes grown until
    the photon loss is below loss_fraction, evolve.py:190-194) -- still evaluated on the GPU, through the
    libc2ray-compatible entry points, with host arrays as in the reference.

    Returns (xh_new, phi_ion): end-of-step ionised fraction (laid out like `xh`) and the summed
    photo-ionisation rate (C-ordered with use_gpu=True, Fortran-ordered with use_gpu=False, as in the
    reference, evolve.py:178,200,244-245).
    """
    if not use_gpu:
        return _evolve_cpu_semantics(dt, dr, src_flux, src_pos, max_subbox, subboxsize, loss_fraction, temp, ndens,
                                     xh, photo_thin_table, photo_thick_table, minlogtau, dlogtau, R_max_LLS,
                                     convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet)
    return _evolve(dt, dr, src_flux, src_pos, use_gpu, temp, ndens, xh, photo_thin_table, minlogtau, dlogtau,
                   R_max_LLS, convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet)


def evolve3D_MPI(dt, dr,
                 src_flux, src_pos,
                 use_gpu, max_subbox, subboxsize, loss_fraction,
                 use_mpi, comm, rank, nprocs,
                 temp, ndens, xh,
                 photo_thin_table, photo_thick_table,
                 minlogtau, dlogtau,
                 R_max_LLS, convergence_fraction,
                 sig, bh00, albpow, colh0, temph0, abu_c,
                 logfile="pyC2Ray.log", quiet=False):
    """Source-sharded variant (pyc2ray/evolve.py:249-498): rank r traces the contiguous block
    [r*(Ns//nprocs), (r+1)*(Ns//nprocs)) of the source list, the last rank to the end
    (evolve.py:362-367); the per-rank rate grids are summed across ranks each iteration.

    `use_mpi`/`comm` may be mpi4py's ``MPI`` module and a communicator, exactly as the reference takes
    them (host-staged Reduce+Bcast), or ``pyc2ray_amd.dist.MPI`` and a ``pyc2ray_amd.dist.TorchComm``
    (one process per GPU under torch.distributed: RCCL all-reduce over xGMI directly on the
    device-resident grid).  All ranks return the same (xh_new, phi_ion).
    """
    if not use_gpu:
        return _evolve_cpu_semantics(dt, dr, src_flux, src_pos, max_subbox, subboxsize, loss_fraction, temp, ndens,
                                     xh, photo_thin_table, photo_thick_table, minlogtau, dlogtau, R_max_LLS,
                                     convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet,
                                     use_mpi=use_mpi, comm=comm, rank=rank, nprocs=nprocs)
    return _evolve(dt, dr, src_flux, src_pos, use_gpu, temp, ndens, xh, photo_thin_table, minlogtau, dlogtau,
                   R_max_LLS, convergence_fraction, sig, bh00, albpow, colh0, temph0, abu_c, logfile, quiet,
                   use_mpi=use_mpi, comm=comm, rank=rank, nprocs=nprocs)
