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
@pytest.mark.parametrize("mode", ["real_plain", "real_overlap", "real_cpu_semantics"])
def test_two_ranks_on_one_gpu_with_the_hip_library(tmp_path, mode):
    """The same two-rank runs with the HIP library itself under both ranks (both on GPU 0, sums staged through the
    host by gloo): the sharded evolve3D_MPI -- plain, pipelined, and with use_gpu=False semantics -- against the
    single-process oracle loop."""
    test_two_ranks_match_single_process(tmp_path, mode)


@pytest.mark.parametrize("mode", ["plain", "overlap", "cpu_semantics"])
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
