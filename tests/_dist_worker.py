"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on the CPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _shutdown():
    """Take the process group down in an orderly way (gloo's threads otherwise race the interpreter's exit)."""
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    overlap = len(sys.argv) > 5 and sys.argv[5] == "overlap"
    cpu_semantics = len(sys.argv) > 5 and sys.argv[5] in ("cpu_semantics", "real_cpu_semantics")
    real = len(sys.argv) > 5 and sys.argv[5].startswith("real_")      # the HIP library itself, every rank on GPU 0
    overlap = overlap or (len(sys.argv) > 5 and sys.argv[5] == "real_overlap")
    mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
    # slab modes: "slab:N:ns:R[:real][:mpi]" -- rates exchanged plane-wise (pyc2ray_amd.dist.SlabPlan)
    spec = mode.split(":")
    slab = spec[0] == "slab"
    real = real or "real" in spec[1:]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import cases
    from fake_backend import OracleAsora, OracleC2Ray
    import pyc2ray_amd.evolve as ev
    from pyc2ray_amd import dist as pd

    r, w, _ = pd.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    comm = pd.TorchComm(overlap=overlap, chunks=4, pipeline_chemistry=overlap)
    comm.exchange = "slab" if slab else "allreduce"
    # "..._legacy" / ":legacy": the all-reduce exchange with three calls and a host read-back per iteration instead of the device loop
    comm.device_loop = not (mode.endswith("_legacy") or "legacy" in spec[1:])
    for item in spec[1:]:
        if item.startswith("k") and item[1:].isdigit():
            comm.slab_chunks = int(item[1:])              # trace chunks of the overlapped slab exchange
    assert comm.Get_rank() == rank and comm.Get_size() == world

    # mpi4py-flavoured surface
    buf = np.full(5, float(rank + 1))
    comm.Allreduce(pd.MPI.IN_PLACE, [buf, pd.MPI.DOUBLE], op=pd.MPI.SUM)
    assert np.all(buf == sum(range(1, world + 1)))
    b2 = np.arange(4.0) if rank == 0 else np.zeros(4)
    comm.Bcast([b2, pd.MPI.DOUBLE], root=0)
    assert np.array_equal(b2, np.arange(4.0))
    b3 = np.full(3, float(rank + 1))
    if rank == 0:
        comm.Reduce(pd.MPI.IN_PLACE, [b3, pd.MPI.DOUBLE], op=pd.MPI.SUM, root=0)
        assert np.all(b3 == sum(range(1, world + 1)))
    else:
        comm.Reduce([b3, pd.MPI.DOUBLE], None, op=pd.MPI.SUM, root=0)

    # the sharded evolve step on the oracle-backed stand-in
    N, ns, R, dt = 16, 5, 6.0, 3.15576e13 * 5
    if slab:
        N, ns, R = int(spec[1]), int(spec[2]), float(spec[3])
    nd, xh, dr = cases.grid(N, "lognormal", 51, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, ns, 52, flux=30.0)       # 5 sources over 2 ranks: 2 + 3
    if slab:
        flux = flux * (3e-4 * (N / 16.0) ** 3 / ns / 30.0) * (1.0 + 0.1 * np.arange(ns))    # a partially ionised box, unequal fluxes
    thin, thick, dlog = cases.soft_tables()
    use_mpi, the_comm = pd.MPI, comm
    if "mpi" in spec[1:]:
        # an mpi4py-shaped communicator (Reduce / Bcast on numpy buffers and nothing else): the host-staged branch of
        # evolve3D_MPI that a real mpi4py communicator takes (pyc2ray/evolve.py:433-437,484-489)
        class _MPIOnly:
            def Reduce(self, sendbuf, recvbuf, op=None, root=0):
                return comm.Reduce(sendbuf, recvbuf, op=op, root=root)

            def Bcast(self, buf, root=0):
                return comm.Bcast(buf, root=root)
        the_comm = _MPIOnly()
    if spec[0] == "cfg3":
        # BASELINE configs[3] at full size, sharded over the ranks (all of them on GPU 0, planes staged through the host by
        # gloo): ONE slab iteration of the real library -- trace of the rank's sources, rates to the owners, slab chemistry,
        # xh_av back -- then the owners' slabs gathered; the caller compares with the reference Fortran's fixture
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import bench
        import make_bigconfig_golden as MB
        import pyc2ray_amd as p
        from pyc2ray_amd import _capi
        from pyc2ray_amd.load_extensions import load_asora
        from pyc2ray_amd.utils.sourceutils import format_sources
        N, NS, R = 256, 1000, 32.0
        thin, thick, dlog = bench.make_tables()
        if "512" in spec[1:]:       # configs[4]'s grid and source list, the 256 sources of the reference fixture (cfg3:512)
            N = 512
            ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, 100000)
            sub = MB.subset_indices(100000, 256)
            pos, flux = pos[:, sub], flux[sub]
            NS = 256
        else:
            ndens, xh, temp, dr, pos, flux = bench.make_workload("cosmo", N, NS)
        lib = load_asora()
        p.device_init(N, 8, device_id=0)
        p.photo_table_to_device(thin, thick)
        comm.exchange = "slab"
        spos, sflux, bounds = comm.shard_sources_by_slab(pos, flux, world)
        plan = pd.SlabPlan(N, world, R, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(world)])
        lo, hi = bounds[rank], bounds[rank + 1]
        p0, f0 = format_sources(spos[:, lo:hi], sflux[lo:hi])
        lib.source_data_to_device(p0, f0, hi - lo)
        for which, a in ((_capi.GRID_NDENS, ndens), (_capi.GRID_TEMP, temp), (_capi.GRID_XH, xh)):
            lib.grid_to_device(which, a)
        lib.grid_copy(_capi.GRID_XH_AV, _capi.GRID_XH)
        lib.grid_copy(_capi.GRID_XH_INTERMED, _capi.GRID_XH)
        lib.set_option(_capi.OPT_FORTRAN_CONSTANTS, 1)          # the fixture is the Fortran path
        chem = (bench.MYR, bench.BH00, bench.ALBPOW, bench.COLH0, bench.TEMPH0, bench.ABU_C)
        conv, s1, s0 = comm.slab_iteration(lib, plan, N, R, bench.SIG, dr, hi - lo, bench.MINLOGTAU, dlog, thin.shape[0] - 1, chem, True)
        comm.slab_gather(lib, plan, _capi.GRID_PHI_ION, N)
        comm.slab_gather(lib, plan, _capi.GRID_XH_INTERMED, N)
        phi = lib.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
        x = lib.grid_to_host(_capi.GRID_XH_INTERMED, np.empty((N, N, N)))
        d = MB.digest(phi)
        src_flat = ((pos[0] - 1) * N + (pos[1] - 1)) * N + (pos[2] - 1)
        np.savez(out, vals=phi.ravel()[MB.sample_indices(N, pos, 20260300 + N)], src_vals=phi.ravel()[src_flat], plane_sums=d["plane_sums"],
                 block_sums=d["block_sums"], nonzero=d["nonzero"], total=d["total"], conv=conv, s1=s1, s0=s0, x_sum=float(x.sum()),
                 nsrc=hi - lo, sent=plan.bytes_per_rank(rank)[0])
        p.device_close()
        comm.Barrier()
        _shutdown()
        return
    golden_case = None
    if spec[0] == "mpigolden":
        # a use_gpu=True case of tests/cases.py, every time step of it: compared with the REFERENCE'S evolve3D_MPI
        # (tests/golden/evolve_mpi.npz) by the caller
        golden_case = cases.evolve_case(spec[1])
        c = golden_case
        N, R, dt, nd, xh, temp, pos, flux, dr = c["N"], c["R"], c["dt"], c["ndens"], c["xh"], c["temp"], c["pos"], c["flux"], c["dr"]
        thin, thick, dlog = c["thin"], c["thick"], c["dlogtau"]
        comm.exchange = "allreduce" if "allreduce" in spec[2:] else "slab"
    fake = OracleAsora(thin, thick)
    if real:
        import pyc2ray_amd as p
        p.device_init(N, 8, device_id=0)
        p.photo_table_to_device(thin, thick)
    else:
        ev.load_asora = lambda: fake
        ev.cuda_is_init = lambda: True
        ev.load_c2ray = lambda: OracleC2Ray()
    if golden_case is not None:
        res = {}
        for step in range(golden_case["steps"]):
            xh, phi = ev.evolve3D_MPI(dt, dr, flux, pos, True, 1000, N, 1e-2, use_mpi, the_comm, rank, world, temp, nd, xh,
                                      thin, thick, cases.MINLOGTAU, dlog, R, golden_case["convergence_fraction"], cases.SIG,
                                      cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C, logfile=None, quiet=True)
            res[f"xh{step}"], res[f"phi{step}"], res[f"niter{step}"] = np.array(xh), np.array(phi), ev._evolve.last_niter
        np.savez(out, **res)
        comm.Barrier()
        _shutdown()
        return
    diag = "phases" in spec[1:]
    if diag:                      # bench.py's diagnostics: where an iteration's time goes, what the links deliver
        comm.phase_timing = True
        if "allreduce" in spec[1:]:
            comm.exchange = "allreduce"
    xh_new, phi = ev.evolve3D_MPI(dt, dr, flux, pos, not cpu_semantics, 1000, 3, 1e-2, use_mpi, the_comm, rank, world,
                                  temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, R, 1e-4, cases.SIG,
                                  cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C,
                                  logfile=None, quiet=True)
    extra = {}
    if diag:
        import json
        if comm.exchange == "allreduce" and not comm.device_loop:
            # evolve3D_MPI makes the three calls itself on this path; bench.py --overlap 1 goes through raytrace_and_allreduce: two steps of it
            lib_ = ev.load_asora()
            NumTau = thin.shape[0]
            lib_.source_data_to_device(*__import__("pyc2ray_amd.utils.sourceutils", fromlist=["format_sources"]).format_sources(pos, flux), ns)
            comm.phase_reset()
            for _ in range(2):
                comm.raytrace_and_allreduce(lib_, N, R, cases.SIG, dr, ns, cases.MINLOGTAU, dlog, NumTau,
                                            chemistry=(dt, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C))
        extra["phases"] = json.dumps(comm.phase_report())
        extra["links"] = json.dumps(comm.measure_links(p2p_bytes=1 << 16, allreduce_bytes=1 << 18, reps=2))
        comm.phase_reset()
        extra["phases_after_reset"] = json.dumps(comm.phase_report())
    np.savez(out, xh=xh_new, phi=phi, niter=ev._evolve.last_niter, nsrc=(fake.flux.shape[0] if fake.flux is not None else -1), **extra)
    comm.Barrier()
    _shutdown()


if __name__ == "__main__":
    main()
