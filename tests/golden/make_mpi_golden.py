"""Golden vectors for evolve3D_MPI, produced by the REFERENCE'S OWN Python code.

Run in the build container (where /root/reference exists):

    make -C oracle
    python tests/golden/make_mpi_golden.py

What runs: /root/reference/pyc2ray/evolve.py::evolve3D_MPI (ref: evolve.py:249-498), loaded where it lies exactly as
tests/golden/make_evolve_golden.py loads evolve3D (same synthetic package, same extension stand-ins: the compiled
reference Fortran for libc2ray, this repository's C restatement of the ASORA kernel for the CUDA module).  mpi4py is not
installed here and there is no mpirun; the function only ever calls `comm.Reduce`, `comm.Bcast` and reads
`use_mpi.IN_PLACE / DOUBLE / SUM` (ref: evolve.py:433-437,480-497), so the P ranks of a run are P THREADS of this process,
each calling the reference's function with its own rank and a communicator object whose Reduce / Bcast meet at a
barrier and sum / copy numpy buffers in rank order -- what MPI_Reduce(SUM) + MPI_Bcast do.  The CUDA stand-in keeps its
state (sources, density) per thread, as every MPI rank has its own GPU context.

Cases (tests/cases.py): the use_gpu=True cases of evolve.npz with nprocs = 1, 2, 3.  Stored per case and P: xh_new,
phi_ion of rank 0, whether all ranks returned identical arrays, the per-iteration convergence rows of rank 0's log, the
number of sources each rank reported.  nprocs = 1 must reproduce evolve3D's fixture bit for bit (asserted here).

(The reference's use_gpu=False branch is not a fixture: there every rank traces ALL sources -- the source split sits
inside `if use_gpu:`, ref: evolve.py:358-371,416-423 -- and Reduce(SUM) multiplies the rates by nprocs.)
"""
import os
import re
import sys
import tempfile
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

import cases  # noqa: E402
import make_evolve_golden as MEG  # noqa: E402
from oracle import ref_fortran as F  # noqa: E402

MPI_CASES = ("l16_gpu_F", "l24_gpu_F_37src")
RANKS = (1, 2, 3)


class ThreadMPI:
    """The attributes of mpi4py.MPI the reference reads."""
    IN_PLACE, DOUBLE, INT, SUM = "IN_PLACE", "DOUBLE", "INT", "SUM"


class ThreadComm:
    """Reduce(SUM) to a root and Bcast between the threads of one process, on the buffers mpi4py would be given."""

    def __init__(self, nprocs):
        self.n = nprocs
        self.barrier = threading.Barrier(nprocs)
        self.slots = [None] * nprocs
        self.tls = threading.local()

    def bind(self, rank):
        self.tls.rank = rank

    @staticmethod
    def _buf(spec):
        return spec[0] if isinstance(spec, (list, tuple)) else spec

    def Reduce(self, sendbuf, recvbuf, op=None, root=0):
        me = self.tls.rank
        mine = self._buf(recvbuf) if isinstance(sendbuf, str) else self._buf(sendbuf)
        self.slots[me] = mine
        self.barrier.wait()
        if me == root:
            acc = np.array(self.slots[0], dtype=np.float64, copy=True)
            for r in range(1, self.n):                       # rank order
                acc += self.slots[r]
            out = self._buf(recvbuf)
            out[...] = acc
        self.barrier.wait()

    def Bcast(self, buf, root=0):
        me = self.tls.rank
        arr = self._buf(buf)
        if me == root:
            self.slots[root] = arr
        self.barrier.wait()
        if me != root:
            src = self.slots[root]
            if isinstance(arr, np.ndarray):
                arr[...] = np.asarray(src).reshape(arr.shape)
            else:                                            # array.array('i', [converged]), ref: evolve.py:484-487
                for q in range(len(arr)):
                    arr[q] = src[q]
        self.barrier.wait()


class PerThreadAsora(MEG._OracleAsora):
    """The CUDA stand-in with one state per thread (one device context per MPI rank)."""
    _tls = threading.local()

    def __getattr__(self, name):
        try:
            return self._tls.__dict__[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self._tls.__dict__[name] = value


def run_case(ev, core, name, nprocs):
    c = cases.evolve_case(name)
    N = c["N"]
    comm = ThreadComm(nprocs)
    results = [None] * nprocs
    errors = []
    with tempfile.TemporaryDirectory() as tmp:
        logs = [os.path.join(tmp, f"log{r}") for r in range(nprocs)]

        def rank_main(rank):
            try:
                comm.bind(rank)
                core.device_init(N, 8)
                core.photo_table_to_device(c["thin"], c["thick"])
                xh = c["xh"]
                outs = []
                for step in range(c["steps"]):
                    xh_new, phi = ev.evolve3D_MPI(c["dt"], c["dr"], c["flux"], c["pos"], True, c["max_subbox"], c["subboxsize"],
                                                  c["loss_fraction"], ThreadMPI, comm, rank, nprocs, c["temp"], c["ndens"], xh,
                                                  c["thin"], c["thick"], cases.MINLOGTAU, c["dlogtau"], c["R"],
                                                  c["convergence_fraction"], cases.SIG, cases.BH00, cases.ALBPOW, cases.COLH0,
                                                  cases.TEMPH0, cases.ABU_C, logfile=logs[rank], quiet=True)
                    outs.append((np.array(xh_new, order="C"), np.array(phi, order="C")))
                    xh = np.asfortranarray(xh_new)
                results[rank] = outs
            except BaseException as e:      # a dead rank would leave the others in the barrier
                errors.append((rank, repr(e)))
                comm.barrier.abort()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nprocs)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        text = open(logs[0]).read()
        nsrc = [int(re.search(r"rank=%d has (\d+) sources" % r, open(logs[r]).read()).group(1)) for r in range(nprocs)]
    rows = MEG.convergence_rows(text)
    return c, results, rows, nsrc


def main():
    assert F.available(), "build oracle/_ref first: make -C oracle"
    ev, rt, core = MEG.load_reference_modules()
    # the reference's modules hold the extension stand-ins as module globals (ref: evolve.py:10-11, asora_core.py:7): one state per thread
    asora = PerThreadAsora()
    ev.libasora = asora
    core.libasora = asora
    single = np.load(os.path.join(HERE, "evolve.npz"))
    out = {}
    for name in MPI_CASES:
        for P in RANKS:
            c, results, rows, nsrc = run_case(ev, core, name, P)
            same = all(np.array_equal(results[r][s][q], results[0][s][q]) for r in range(P) for s in range(c["steps"]) for q in (0, 1))
            for s in range(c["steps"]):
                key = f"{name}__P{P}"
                out[f"{key}__xh{s}"], out[f"{key}__phi{s}"] = results[0][s]
            out[f"{name}__P{P}__rows"] = rows
            out[f"{name}__P{P}__nsrc"] = np.array(nsrc)
            out[f"{name}__P{P}__ranks_identical"] = np.array(same)
            if P == 1:      # one rank: the function must be evolve3D (tests/golden/evolve.npz) bit for bit
                for s in range(c["steps"]):
                    assert np.array_equal(results[0][s][0], single[f"{name}__xh{s}"]), (name, s)
                    assert np.array_equal(results[0][s][1], single[f"{name}__phi{s}"]), (name, s)
                assert len(rows) == sum(len(single[f"{name}__rows{s}"]) for s in range(c["steps"]))
            x_last = results[0][-1][0]
            x_one = out[f"{name}__P1__xh{c['steps'] - 1}"]
            print(f"{name} P={P}: sources per rank {nsrc}, {len(rows)} outer iterations, ranks identical: {same}, "
                  f"<x> = {x_last.mean():.6f}, max |x - x(P=1)| / x = "
                  f"{np.max(np.abs(x_last - x_one) / x_last):.2e}")
    np.savez_compressed(os.path.join(HERE, "evolve_mpi.npz"), **out)
    print("written", os.path.join(HERE, "evolve_mpi.npz"))


if __name__ == "__main__":
    main()
