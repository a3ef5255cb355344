"""CPU: the multi-rank path (source sharding + all-reduce of the rate grid + replicated chemistry)
with world_size 2 over gloo.  The per-rank compute is the oracle-backed stand-in of
tests/fake_backend.py; what is under test is pyc2ray_amd/evolve.py::evolve3D_MPI and
pyc2ray_amd/dist.py::TorchComm."""
import os
import socket
import subprocess
import sys

import numpy as np

import cases
from evolve_oracle import evolve3D_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


import pytest


def test_final_plane_runs_cover_every_plane_once_and_never_early():
    """The slab schedule of the pipelined all-reduce: every plane is released exactly once, and never before
    the last source that can reach it (periodically) has been traced."""
    from pyc2ray_amd.dist import TorchComm
    for N, K, R in ((256, 8, 32.0), (256, 8, 16.0), (48, 5, 9.5), (32, 4, 3.2), (17, 3, 2.0), (16, 8, 40.0), (64, 1, 5.0)):
        reduced = np.zeros(N, dtype=bool)
        reach = int(np.floor(R))
        for c in range(K):
            runs = TorchComm.final_plane_runs(N, K, R, c, reduced)
            for a, b in runs:
                assert 0 <= a < b <= N and not reduced[a:b].any()
                # a source traced later sits at i >= (c+1)*N//K and touches (i + d) mod N, |d| <= floor(R)
                for i in range((c + 1) * N // K, N):
                    touched = {(i + d) % N for d in range(-reach, reach + 1)}
                    assert not touched.intersection(range(a, b)), (N, K, R, c, a, b, i)
                reduced[a:b] = True
        assert reduced.all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["real_plain", "real_plain_legacy", "real_overlap", "real_cpu_semantics"])
def test_two_ranks_on_one_gpu_with_the_hip_library(tmp_path, mode):
    """The same two-rank runs with the HIP library itself under both ranks (both on GPU 0, sums staged through the
    host by gloo): the sharded evolve3D_MPI -- plain, pipelined, and with use_gpu=False semantics -- against the
    single-process oracle loop."""
    test_two_ranks_match_single_process(tmp_path, mode)


@pytest.mark.parametrize("mode", ["plain", "plain_legacy", "overlap", "cpu_semantics"])
def test_two_ranks_match_single_process(tmp_path, mode):
    world = 2
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    env = dict(os.environ, PYC2RAY_AMD_NO_TORCH="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), port,
                               outs[r], mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    res = [np.load(o) for o in outs]
    # contiguous source blocks, last rank takes the remainder (pyc2ray/evolve.py:362-367)
    if not mode.startswith("real_"):
        assert [int(r["nsrc"]) for r in res] == [2, 3]
    # every rank returns the same fields
    assert np.array_equal(res[0]["xh"], res[1]["xh"]) and np.array_equal(res[0]["phi"], res[1]["phi"])
    assert int(res[0]["niter"]) == int(res[1]["niter"])

    N = 16
    nd, xh, dr = cases.grid(N, "lognormal", 51, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, 5, 52, flux=30.0)
    thin, thick, dlog = cases.soft_tables()
    if mode.endswith("cpu_semantics"):
        # use_gpu=False: each rank runs the sub-box raytracer on its block of sources (equal fluxes, so the
        # reference's flux-of-the-last-source convention is immaterial), rates summed over ranks on the host
        from evolve_oracle import evolve3d_cpu_path
        x_ref, phi_ref, niter_ref = evolve3d_cpu_path(3.15576e13 * 5, dr, flux, pos, 1000, 3, 1e-2, temp, nd, xh, thin,
                                                      thick, cases.MINLOGTAU, dlog, 6.0, 1e-4, cases.SIG)
    else:
        x_ref, phi_ref, niter_ref, _ = evolve3D_oracle(3.15576e13 * 5, dr, flux, pos, temp, nd, xh, thin, thick,
                                                       cases.MINLOGTAU, dlog, 6.0, 1e-4, cases.SIG, cases.BH00,
                                                       cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)
    assert int(res[0]["niter"]) == niter_ref
    rtol = 1e-7 if mode.startswith("real_") else 1e-10          # the HIP kernels vs the oracle: tests/test_gpu_parity.py
    np.testing.assert_allclose(res[0]["xh"], x_ref, rtol=rtol, atol=0)
    np.testing.assert_allclose(res[0]["phi"], phi_ref, rtol=rtol, atol=0)


# ---- bench.py --gpus N without a launcher ----------------------------------------------------------
def _bench(args, timeout=240, **env):
    e = dict(os.environ, PYC2RAY_AMD_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` with no launcher around it starts its N ranks itself (a child torch.distributed.run), relays
    rank 0's single JSON line and exits 0.  --launch-check runs the launch path without the measurement: process group up
    over gloo, one sum over the ranks."""
    import json
    r = _bench(["--gpus", str(world), "--launch-check"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line on stdout, whatever the ranks and the launcher print
    d = json.loads(lines[0])
    assert d["launch_check"] is True and d["world_size_reported_by_backend"] == world and d["self_launched"] is True
    assert d["host"]["logical_cpus"] >= 1 and d["host"]["sockets"] >= 1


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that fails ends the run with a non-zero exit code and the child's stderr, well inside the timeout; nothing that
    looks like a result reaches stdout."""
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2", "--launch-check", "1"], PYC2RAY_AMD_DIST_TIMEOUT_S="20")
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "fails on request" in r.stderr and r.stdout.strip() == ""


def test_bench_self_launch_kills_a_child_that_hangs():
    """--launch-timeout bounds the child: rank 1 never joins the group here (it fails on request while rank 0 waits for it
    with a long collective timeout), the launcher is killed with its ranks and the exit code is 124 or the child's own."""
    r = _bench(["--gpus", "2", "--launch-check", "--launch-timeout", "8"], PYC2RAY_AMD_BENCH_TEST_HANG="1")
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert _rank_processes("--launch-timeout 8") == []


def _rank_processes(marker):
    found = []
    for pid in (d for d in os.listdir("/proc") if d.isdigit()):
        try:
            with open(f"/proc/{pid}/cmdline", "rb") as f:
                args = f.read().replace(b"\0", b" ").decode(errors="replace")
        except OSError:
            continue
        if marker in args and "bench.py" in args and int(pid) != os.getpid():
            found.append(int(pid))
    return found


def test_bench_self_launch_takes_its_ranks_along_when_it_is_terminated():
    """Whoever started the self-launching bench.py gives up (SIGTERM, as a driver's timeout sends): the ranks -- each a session of
    its own under torch.distributed.run, out of reach of a signal to the launcher's group -- are ended with it instead of being
    left on the GPUs.  The same holds for --launch-timeout (the test above): no rank is left behind either way."""
    import signal
    import time
    marker = "--launch-timeout 201"          # (recognises this test's processes among everything else that runs)
    e = dict(os.environ, PYC2RAY_AMD_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", PYC2RAY_AMD_BENCH_TEST_HANG="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--launch-check"] + marker.split(),
                         env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 90
        while time.time() < deadline and len(_rank_processes(marker)) < 4:        # this process, the launcher, two ranks
            time.sleep(0.5)
        assert len(_rank_processes(marker)) >= 4, "the ranks never started"
        time.sleep(2.0)
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=40) == 128 + signal.SIGTERM
        time.sleep(1.0)
        assert _rank_processes(marker) == []
    finally:
        if p.poll() is None:
            p.kill()
        for pid in _rank_processes(marker):
            os.kill(pid, signal.SIGKILL)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench(["--gpus", "2", "--launch-check"], WORLD_SIZE="3", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


# ---- slab exchange (pyc2ray_amd.dist.SlabPlan): rates to the owners of the planes, slab chemistry, xh_av back -------
def test_slab_plan_covers_what_the_sources_reach():
    """Pure bookkeeping: for random source sets, radii and rank counts, every plane a rank's sources can rate lies in
    its reach, every reached plane of a foreign slab lies in the run sent to that slab's owner, the slabs partition
    the planes, and the byte counts are symmetric between the two exchanges."""
    from pyc2ray_amd.dist import SlabPlan, TorchComm
    rng = np.random.default_rng(7)
    for trial in range(60):
        N = int(rng.choice([16, 17, 24, 33, 64, 256]))
        P = int(rng.choice([1, 2, 3, 4, 8]))
        R = float(rng.choice([0.0, 1.5, 3.0, N / 8.0, N / 4.0 + 0.5, N / 2.0, 10.0 * N]))
        ns = int(rng.integers(P, 6 * P + 1))
        pos = 1 + rng.integers(0, N, size=(3, ns))
        if trial % 3 == 0:
            pos[0] = 1 + (pos[0] % max(2, N // 5))                  # all sources in a thin slab near the periodic seam
        flux = rng.uniform(0.5, 2.0, size=ns)
        spos, sflux, bounds = TorchComm.shard_sources_by_slab(pos, flux, P)
        assert sorted(map(tuple, spos.T)) == sorted(map(tuple, pos.T)) and np.all(np.diff(spos[0]) >= 0)
        assert bounds[0] == 0 and bounds[-1] == ns and all(bounds[r + 1] - bounds[r] >= ns // P for r in range(P))
        plan = SlabPlan(N, P, R, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(P)])
        owner = np.full(N, -1)
        for q, (a, b) in enumerate(plan.own):
            assert np.all(owner[a:b] == -1)
            owner[a:b] = q
        assert np.all(owner >= 0)
        m = int(np.floor(R))
        for r in range(P):
            want = np.zeros(N, dtype=bool)
            for i0 in spos[0, bounds[r]:bounds[r + 1]] - 1:
                for d in range(-min(m, N // 2), min(m, N // 2 - 1 + N % 2) + 1):     # raytracing.cu:122-123
                    want[(i0 + d) % N] = True
            assert np.array_equal(want, plan.reach[r])
            for q in range(P):
                a, b = plan.own[q]
                idx = np.flatnonzero(want[a:b])
                runs = plan.runs[r][q]
                if idx.size == 0:
                    assert runs == []
                else:
                    covered = np.zeros(N, dtype=bool)
                    for ra, rb in runs:
                        assert a <= ra < rb <= b and not covered[ra:rb].any()
                        covered[ra:rb] = True
                    assert covered[a:b][want[a:b]].all()                         # every reached plane travels
                    assert want[[ra for ra, _ in runs]].all() and want[[rb - 1 for _, rb in runs]].all()   # runs start and end on reached planes
                    assert len(runs) <= 1 + (want[a:b].size - int(want[a:b].sum())) // SlabPlan.MERGE_GAP
            # the chunked schedule: every plane of every run leaves exactly once, and never while a later chunk can still reach it
            for K in (1, 3, 4):
                sched = plan.send_schedule(r, K)
                assert len(sched) == K
                bnd = plan.chunk_bounds(bounds[r + 1] - bounds[r], K)
                mine = spos[0, bounds[r]:bounds[r + 1]] - 1
                gone = np.zeros(N, dtype=int)
                for c, pieces in enumerate(sched):
                    later = np.zeros(N, dtype=bool)
                    for i0 in mine[bnd[c + 1]:]:
                        for d in range(-min(m, N // 2), min(m, N // 2 - 1 + N % 2) + 1):
                            later[(i0 + d) % N] = True
                    for q, pa, pb in pieces:
                        assert q != r and plan.own[q][0] <= pa < pb <= plan.own[q][1]
                        assert not later[pa:pb].any(), (trial, r, K, c)
                        gone[pa:pb] += 1
                want_gone = np.zeros(N, dtype=int)
                for q in range(P):
                    if q != r:
                        for ra, rb in plan.runs[r][q]:
                            want_gone[ra:rb] += 1
                assert np.array_equal(gone, want_gone)
                # what q expects from r is what r sends to q, chunk by chunk
                for q in range(P):
                    if q == r:
                        continue
                    rs = plan.recv_schedule(q, K)
                    for c in range(K):
                        assert [(pa, pb) for (src, pa, pb) in rs[c] if src == r] == [(pa, pb) for (dest, pa, pb) in sched[c] if dest == q]
            work = np.zeros(N, dtype=bool)
            for a, b in plan.work_runs(r):
                work[a:b] = True
            assert np.array_equal(work, want | (owner == r))
        sent = sum(plan.bytes_per_rank(r)[0] for r in range(P))
        recv = sum(plan.bytes_per_rank(r)[1] for r in range(P))
        assert sent == recv
    # the benchmark configuration: 256^3, R = 32, 8 ranks, evenly spread sources -> two neighbours, 32 planes each
    pos = np.stack([1 + np.arange(1000) % 256, np.ones(1000, int), np.ones(1000, int)])
    spos, _, bounds = TorchComm.shard_sources_by_slab(pos, np.ones(1000), 8)
    plan = SlabPlan(256, 8, 32.0, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(8)])
    per_exchange = max(plan.bytes_per_rank(r)[0] for r in range(8))
    assert per_exchange <= 70 * 256 * 256 * 8                       # ~2 x 32 planes of 512 KiB, against 2 x 7/8 x 128 MiB in a ring all-reduce
    # two ranks: a rank reaches the other's slab from both sides around the periodic box -- two runs of ~32 planes, not the whole slab
    spos, _, bounds = TorchComm.shard_sources_by_slab(pos, np.ones(1000), 2)
    plan2 = SlabPlan(256, 2, 32.0, [spos[0, bounds[r]:bounds[r + 1]] - 1 for r in range(2)])
    assert [len(plan2.runs[r][1 - r]) for r in range(2)] == [2, 2]
    assert max(plan2.bytes_per_rank(r)[0] for r in range(2)) <= 68 * 256 * 256 * 8           # (64 MiB per exchange with one covering run)


def _run_workers(tmp_path, world, mode, timeout=600):
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    env = dict(os.environ, PYC2RAY_AMD_NO_TORCH="0", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), port,
                               outs[r], mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [np.load(o) for o in outs]


def _slab_reference(N, ns, R):
    nd, xh, dr = cases.grid(N, "lognormal", 51, 0.15, xlo=1e-4, xhi=2e-3)
    temp = np.full((N, N, N), 1e4)
    pos, flux = cases.sources(N, ns, 52, flux=30.0)
    flux = flux * (3e-4 * (N / 16.0) ** 3 / ns / 30.0) * (1.0 + 0.1 * np.arange(ns))
    thin, thick, dlog = cases.soft_tables()
    return evolve3D_oracle(3.15576e13 * 5, dr, flux, pos, temp, nd, xh, thin, thick, cases.MINLOGTAU, dlog, R, 1e-4,
                           cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0, cases.TEMPH0, cases.ABU_C)


@pytest.mark.parametrize("world,N,ns,R,chunks", [(2, 24, 13, 3.0, 3), (3, 33, 17, 4.0, 4), (4, 24, 21, 2.5, 2), (8, 32, 41, 2.0, 3),
                                                 (2, 16, 9, 1000.0, 3), (4, 24, 5, 3.0, 4)])
def test_overlapped_slab_exchange_matches_single_process(tmp_path, world, N, ns, R, chunks):
    """The slab exchange with the trace in K chunks and every foreign plane sent as soon as no later chunk can reach it
    (TorchComm.slab_chunks; SlabPlan.send_schedule).  A plane sent too early would miss the rates of a later chunk, a
    plane folded twice trips the stand-in's assertion, a receive nobody sends hangs the job: same fields and iteration
    count as the single-process oracle loop.  Includes a radius beyond the box (everything final only at the end) and
    ranks with fewer sources than chunks (empty chunks)."""
    res = _run_workers(tmp_path, world, f"slab:{N}:{ns}:{R}:k{chunks}")
    for r in res[1:]:
        assert np.array_equal(r["xh"], res[0]["xh"]) and np.array_equal(r["phi"], res[0]["phi"])
        assert int(r["niter"]) == int(res[0]["niter"])
    x_ref, phi_ref, niter_ref, _ = _slab_reference(N, ns, R)
    assert int(res[0]["niter"]) == niter_ref and niter_ref >= 3
    np.testing.assert_allclose(res[0]["xh"], x_ref, rtol=1e-10, atol=0)
    np.testing.assert_allclose(res[0]["phi"], phi_ref, rtol=1e-10, atol=0)


@pytest.mark.parametrize("world,N,ns,R", [(2, 16, 5, 6.0), (3, 17, 7, 2.5), (4, 24, 9, 3.0), (8, 16, 19, 2.0), (4, 16, 6, 1000.0)])
def test_slab_exchange_matches_single_process(tmp_path, world, N, ns, R):
    """evolve3D_MPI with the rates exchanged plane-wise (the default with a TorchComm), 2 to 8 ranks over gloo,
    against the single-process oracle loop: same iteration count, same fields on every rank.  Radii small against
    the slabs (some pairs of ranks exchange nothing), odd meshes, uneven slabs, and R beyond the box (every rank
    reaches every plane).  The per-rank compute is the oracle-backed stand-in, which leaves NaN wherever a plan
    would forget to form nHI or to zero the accumulator."""
    res = _run_workers(tmp_path, world, f"slab:{N}:{ns}:{R}")
    for r in res[1:]:
        assert np.array_equal(r["xh"], res[0]["xh"]) and np.array_equal(r["phi"], res[0]["phi"])
        assert int(r["niter"]) == int(res[0]["niter"])
    assert sum(int(r["nsrc"]) for r in res) == ns
    x_ref, phi_ref, niter_ref, _ = _slab_reference(N, ns, R)
    assert int(res[0]["niter"]) == niter_ref and niter_ref >= 3
    assert 0.02 < x_ref.mean() < 0.98
    np.testing.assert_allclose(res[0]["xh"], x_ref, rtol=1e-10, atol=0)
    np.testing.assert_allclose(res[0]["phi"], phi_ref, rtol=1e-10, atol=0)


@pytest.mark.parametrize("world,exchange", [(2, "slab"), (3, "slab"), (2, "allreduce"), (3, "allreduce"), (2, "allreduce:legacy")])
def test_phase_times_and_link_rates_of_a_multi_rank_run(tmp_path, world, exchange):
    """What bench.py --gpus N prints beside its value: per-phase milliseconds of an iteration -- booked by
    TorchComm.slab_enqueue / raytrace_and_allreduce when `phase_timing` is set, maximum over the ranks -- and the rates a
    point-to-point ring and an all-reduce reach in this job (TorchComm.measure_links).  Every rank reports the same numbers
    (they drive `--exchange auto`, which all ranks must decide alike); switching the timers on changes no result."""
    import json
    res = _run_workers(tmp_path, world, f"slab:16:5:6.0:phases" + (":" + exchange if exchange != "slab" else ""))
    x_ref, phi_ref, niter_ref, _ = _slab_reference(16, 5, 6.0)
    for r in res:
        assert int(r["niter"]) == niter_ref
        np.testing.assert_allclose(r["xh"], x_ref, rtol=1e-10, atol=0)
    ph = [json.loads(str(r["phases"])) for r in res]
    ln = [json.loads(str(r["links"])) for r in res]
    assert all(q == ph[0] for q in ph[1:]) and all(q == ln[0] for q in ln[1:])
    want = ({"trace_fold_post", "wait_rates_add", "slab_pass", "xh_av_exchange_nhi", "scalar_allreduce_test"}
            if exchange == "slab" else {"trace_fold", "rate_allreduce", "pass_test"} if exchange == "allreduce"     # (the device loop)
            else {"trace", "rate_allreduce", "chemistry"})
    assert set(ph[0]) == want | {"iterations"}
    assert ph[0]["iterations"] == (2 if exchange.endswith("legacy") else niter_ref)
    assert all(ph[0][k] >= 0.0 for k in want) and sum(ph[0][k] for k in want) > 0.0
    for key in ("p2p_ms", "p2p_GBs", "allreduce_ms", "allreduce_busbw_GBs"):
        assert ln[0][key] > 0.0
    assert ln[0]["ranks"] == world and ln[0]["p2p_bytes"] == 1 << 16 and ln[0]["allreduce_bytes"] == 1 << 18
    assert json.loads(str(res[0]["phases_after_reset"])) == {"iterations": 0}


def test_mpi4py_shaped_communicator_takes_the_host_staged_branch(tmp_path):
    """A communicator that only offers mpi4py's Reduce / Bcast on numpy buffers (what evolve3D_MPI gets from a real
    mpi4py run; mpi4py itself is not installed here): Reduce to rank 0 + Bcast of the rate grid, Bcast of the
    convergence flag, as the reference does."""
    res = _run_workers(tmp_path, 2, "slab:16:5:6.0:mpi")
    assert np.array_equal(res[0]["xh"], res[1]["xh"]) and np.array_equal(res[0]["phi"], res[1]["phi"])
    x_ref, phi_ref, niter_ref, _ = _slab_reference(16, 5, 6.0)
    assert int(res[0]["niter"]) == niter_ref
    np.testing.assert_allclose(res[0]["xh"], x_ref, rtol=1e-10, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,ns,R,chunks", [(2, 16, 5, 6.0, 1), (4, 24, 9, 3.0, 1), (2, 24, 13, 3.0, 3), (4, 32, 41, 2.5, 4)])
def test_slab_exchange_with_the_hip_library(tmp_path, world, N, ns, R, chunks):
    """The same with the HIP library under every rank (all on GPU 0, planes staged through the host by gloo), with and
    without the chunked, overlapped first exchange."""
    res = _run_workers(tmp_path, world, f"slab:{N}:{ns}:{R}:real:k{chunks}")
    for r in res[1:]:
        assert np.array_equal(r["xh"], res[0]["xh"]) and int(r["niter"]) == int(res[0]["niter"])
    x_ref, phi_ref, niter_ref, _ = _slab_reference(N, ns, R)
    assert int(res[0]["niter"]) == niter_ref
    np.testing.assert_allclose(res[0]["xh"], x_ref, rtol=1e-7, atol=0)
    np.testing.assert_allclose(res[0]["phi"], phi_ref, rtol=1e-7, atol=0)


# ---- evolve3D_MPI against the REFERENCE'S OWN evolve3D_MPI (tests/golden/make_mpi_golden.py) ---------------------------
def _check_against_reference_mpi(res, name, world, rtol_x, rtol_phi):
    g = np.load(os.path.join(HERE, "golden", "evolve_mpi.npz"))
    c = cases.evolve_case(name)
    assert bool(g[f"{name}__P{world}__ranks_identical"])
    done = 0
    for step in range(c["steps"]):
        for r in res[1:]:
            assert np.array_equal(r[f"xh{step}"], res[0][f"xh{step}"]) and np.array_equal(r[f"phi{step}"], res[0][f"phi{step}"])
        niter = int(res[0][f"niter{step}"])
        done += niter
        np.testing.assert_allclose(res[0][f"xh{step}"], g[f"{name}__P{world}__xh{step}"], rtol=rtol_x, atol=0)
        want = g[f"{name}__P{world}__phi{step}"]
        w = want != 0
        assert np.array_equal(res[0][f"phi{step}"] != 0, w)
        np.testing.assert_allclose(res[0][f"phi{step}"][w], want[w], rtol=rtol_phi, atol=0)
    assert done == len(g[f"{name}__P{world}__rows"])        # outer iterations over all time steps, from the reference's log


@pytest.mark.parametrize("world,name,exchange", [(2, "l16_gpu_F", "slab"), (3, "l16_gpu_F", "allreduce"), (2, "l16_gpu_F", "allreduce:legacy"),
                                                 (2, "l24_gpu_F_37src", "allreduce"), (3, "l24_gpu_F_37src", "slab")])
def test_evolve3D_MPI_reproduces_the_reference_evolve3D_MPI(tmp_path, world, name, exchange):
    """The fixture is the reference's own evolve3D_MPI (pyc2ray/evolve.py:249-498) run with 1, 2 and 3 ranks over its
    Reduce / Bcast calls.  Here: this package's evolve3D_MPI over gloo (oracle-backed compute), slab exchange and full-grid
    all-reduce: same number of outer iterations in every time step, same fields on every rank."""
    res = _run_workers(tmp_path, world, f"mpigolden:{name}:{exchange}")
    _check_against_reference_mpi(res, name, world, 1e-10, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("world,name,exchange", [(2, "l24_gpu_F_37src", "slab"), (3, "l16_gpu_F", "slab"), (2, "l24_gpu_F_37src", "allreduce"),
                                                 (3, "l16_gpu_F", "allreduce")])
def test_evolve3D_MPI_with_the_hip_library_reproduces_the_reference_evolve3D_MPI(tmp_path, world, name, exchange):
    """The same with the HIP library under every rank (all ranks on GPU 0, exchanges staged through the host by gloo): the slab
    exchange and the full-grid all-reduce, both on the device-resident loop."""
    res = _run_workers(tmp_path, world, f"mpigolden:{name}:{exchange}:real")
    _check_against_reference_mpi(res, name, world, 1e-8, 1e-7)


@pytest.mark.gpu
def test_config4_grid_sharded_over_two_ranks_against_reference_fortran(tmp_path):
    """The same at configs[4]'s size: 512^3 (1 GiB grids, plane runs of 64 MiB and more through the exchange), the 256 sources
    of the reference fixture sharded over two ranks."""
    res = _run_workers(tmp_path, 2, "cfg3:512", timeout=1100)
    g = np.load(os.path.join(HERE, "golden", "fullsize_cosmo512_R32.npz"))
    assert sum(int(r["nsrc"]) for r in res) == 256
    for r in res:
        assert int(r["conv"]) == int(res[0]["conv"]) and float(r["x_sum"]) == float(res[0]["x_sum"])
        np.testing.assert_allclose(r["vals"], g["vals"], rtol=1e-8, atol=0)
        np.testing.assert_allclose(r["src_vals"], g["src_vals"], rtol=1e-8, atol=0)
        assert int(r["nonzero"]) == int(g["nonzero"])
        np.testing.assert_allclose(r["plane_sums"], g["plane_sums"], rtol=1e-9)
        np.testing.assert_allclose(float(r["total"]), float(g["total"]), rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_config3_full_size_sharded_over_ranks_against_reference_fortran(tmp_path, world):
    """BASELINE configs[3] at its full size in its multi-GPU mode -- 256^3 log-normal density, the 1000 sources sharded over
    the ranks by slab of the first coordinate, rates sent plane-wise to the owners, slab chemistry, xh_av back -- with the HIP
    library under every rank (all ranks on GPU 0, planes through the host by gloo).  The summed rates every rank ends up with
    against the reference Fortran's sparse fixture (tests/golden/make_bigconfig_golden.py); the slab chemistry's reductions
    identical on every rank."""
    res = _run_workers(tmp_path, world, "cfg3", timeout=900)
    g = np.load(os.path.join(HERE, "golden", "fullsize_cosmo256_R32.npz"))
    assert sum(int(r["nsrc"]) for r in res) == 1000
    for r in res:
        assert int(r["conv"]) == int(res[0]["conv"]) and float(r["s1"]) == float(res[0]["s1"]) and float(r["x_sum"]) == float(res[0]["x_sum"])
        np.testing.assert_allclose(r["vals"], g["vals"], rtol=1e-8, atol=0)
        np.testing.assert_allclose(r["src_vals"], g["src_vals"], rtol=1e-8, atol=0)
        assert int(r["nonzero"]) == int(g["nonzero"])
        np.testing.assert_allclose(r["plane_sums"], g["plane_sums"], rtol=1e-9)
        np.testing.assert_allclose(r["block_sums"], g["block_sums"], rtol=1e-9, atol=1e-12 * float(np.abs(g["block_sums"]).max()))
        np.testing.assert_allclose(float(r["total"]), float(g["total"]), rtol=1e-10)
    assert max(int(r["sent"]) for r in res) < 100 * 256 * 256 * 8          # planes, not grids, travel
