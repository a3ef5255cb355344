"""Multi-GPU communicator: one process per MI355X under torch.distributed.

The reference shards sources over MPI ranks with mpi4py and moves N^3 float64 grids through host
memory (pyc2ray/evolve.py:433-437,480-497).  The path has exactly one real exchange step per outer
iteration -- the sum of the per-rank photo-ionisation rate grids -- so that is the only collective
here: an all-reduce (backend "nccl" == RCCL over xGMI) applied IN PLACE to the device-resident
phi_ion grid.  With the "gloo" backend (CPU tests) the grid is staged through the host.

``TorchComm.raytrace_and_allreduce`` optionally PIPELINES that all-reduce with the raytrace
(``PYC2RAY_AMD_OVERLAP=1`` or ``TorchComm(overlap=True)``): with a rank's sources sorted by their first
coordinate and traced in chunks, the planes of the rate grid that later chunks can no longer reach are final
on every rank, so their sum over ranks runs on a second stream while the next chunk is being traced; only the
planes the last chunks touch are summed after the trace.

``TorchComm`` also offers the handful of mpi4py-style methods the reference's evolve3D_MPI calls
(Get_rank, Get_size, Reduce, Bcast, Allreduce, Barrier), and ``MPI`` the constants it reads, so
reference-style driver code can pass ``use_mpi=dist.MPI, comm=dist.TorchComm()``.
"""
import os

import numpy as np

__all__ = ["MPI", "TorchComm", "SlabPlan", "init_process_group_from_env"]


class _MPIShim:
    """The attributes of mpi4py.MPI that pyc2ray/evolve.py touches."""
    IN_PLACE = "IN_PLACE"
    DOUBLE = "DOUBLE"
    INT = "INT"
    SUM = "SUM"

    @property
    def COMM_WORLD(self):
        return TorchComm()


MPI = _MPIShim()


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  Backend defaults to nccl when a GPU is visible."""
    # the host driver of these machines only supports dmabuf IPC: without this, RCCL's buffer registration across
    # processes fails with "hipIpcGetMemHandle: invalid argument"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        # a collective that never completes aborts the job after this long instead of the backend's default of ten minutes
        # and more (PYC2RAY_AMD_DIST_TIMEOUT_S)
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("PYC2RAY_AMD_DIST_TIMEOUT_S", "300")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local_rank



def _runs_of(mask):
    """Maximal runs of True in a boolean vector as [(begin, end)), ...]."""
    m = np.concatenate(([False], np.asarray(mask, dtype=bool), [False]))
    edges = np.flatnonzero(m[1:] != m[:-1])
    return [(int(a), int(b)) for a, b in zip(edges[0::2], edges[1::2])]


class SlabPlan:
    """Which planes of the grids travel between which ranks in one outer iteration of a multi-GPU step.

    The rate grid is a sum over sources and the sources are sharded (pyc2ray/evolve.py:360-371), so the reference
    sums N^3 grids over ranks every iteration (Reduce + Bcast, evolve.py:433-437) and broadcasts two more after
    the chemistry (evolve.py:480-481).  A source at plane i0 only rates the planes i0-R..i0+R.  With the source list
    ordered by the first coordinate before it is cut into the reference's contiguous blocks, rank r's rates are
    non-zero only on `reach[r]` = the planes within R of its slab of sources.  So:

      * the chemistry of plane i is done by ONE rank, the owner of the slab own[q] = [q N/P, (q+1) N/P) it lies in;
      * exchange 1 (rates): rank r sends the part of reach[r] inside own[q] to q, for every q != r where that is
        not empty; q adds what it receives, in rank order, to its own contribution: the summed rates of own[q];
      * slab chemistry on own[q] (1/P of the grid per rank instead of all of it on every rank);
      * exchange 2 (ionised fraction): the same runs travel back -- q sends the new xh_av of reach[r] & own[q] to r,
        which needs it to form nHI for its next raytrace;
      * the three convergence scalars are summed over ranks in rank order (identical on every rank).

    Planes travel as contiguous runs (plane i is N*N consecutive doubles of the [i][j][k] grid); for a pair (r, q)
    these are the maximal runs of reach[r] inside own[q], runs closer than MERGE_GAP planes joined (zeros in between are
    harmless, every run is a message) -- usually one, two when r's sources reach q's slab from both sides around the
    periodic box (two ranks: always).  At 256^3, R = 32, 8 ranks with evenly spread sources a rank sends and receives
    2 x 32 planes per exchange (2 x 16.8 MiB in, the same out, to and from its two neighbours over their direct xGMI
    links) instead of taking part in a ring all-reduce of 128 MiB (224 MiB in and out per rank), and runs 1/8 of the
    chemistry.  With R >= N/2 every rank reaches every plane and the scheme degenerates into reduce-scatter +
    all-gather by direct sends.

    Overlap (``send_schedule``): a rank traces its sources -- ordered by first coordinate -- in K chunks; a foreign plane
    that no LATER chunk can reach is final, is folded and sent at once, while the next chunk is being traced.  Which
    planes go after which chunk is again a function of the plan only, so the receiver posts the matching receives
    without being told.

    Everything here is a function of (N, P, R, first coordinates of the sorted sources) only, so every rank derives
    the same plan without communicating."""

    #: runs of reached planes closer than this are sent as one run
    MERGE_GAP = 4

    def __init__(self, N, nprocs, R, shard_i0):
        """shard_i0[r]: 0-based first coordinates of the sources of rank r, in the order they are uploaded (ascending
        when the chunked schedule is used)."""
        self.N, self.P = int(N), int(nprocs)
        N, P = self.N, self.P
        self.own = [(q * N // P, (q + 1) * N // P) for q in range(P)]
        m = int(np.floor(R)) if np.isfinite(R) else N
        self._lo, self._hi = min(m, N // 2), min(m, N // 2 - 1 + N % 2)          # the periodic window, raytracing.cu:122-123
        self.i0 = [np.asarray(shard_i0[r], dtype=np.int64) for r in range(P)]
        self.reach = [self._reach_of(self.i0[r]) for r in range(P)]
        # runs[r][q]: runs of planes rank r contributes to (and needs back from) the slab of rank q
        self.runs = [[[] for _ in range(P)] for _ in range(P)]
        for r in range(P):
            for q in range(P):
                a, b = self.own[q]
                merged = []
                for s0, s1 in _runs_of(self.reach[r][a:b]):
                    if merged and s0 - merged[-1][1] < self.MERGE_GAP:
                        merged[-1] = (merged[-1][0], s1)
                    else:
                        merged.append((s0, s1))
                self.runs[r][q] = [(a + s0, a + s1) for s0, s1 in merged]

    def _reach_of(self, i0):
        """Planes the sources at first coordinates i0 can rate."""
        mask = np.zeros(self.N, dtype=bool)
        i0 = np.unique(np.asarray(i0, dtype=np.int64))
        if i0.size:
            if self._lo + self._hi + 1 >= self.N:
                mask[:] = True
            else:
                for d in range(-self._lo, self._hi + 1):
                    mask[(i0 + d) % self.N] = True
        return mask

    def common_chunks(self, K):
        """The chunk count every rank uses: K when every rank's sources are in ascending order of their first coordinate
        (a rank with fewer sources than chunks simply has empty chunks), else 1 -- nothing is final before the end then."""
        if any(np.any(np.diff(i0) < 0) for i0 in self.i0):
            return 1
        return max(1, int(K))

    @staticmethod
    def chunk_bounds(n, K):
        """Rank-local source ranges [b[c], b[c+1]) of the K trace chunks (equal counts, upload order)."""
        return [c * n // K for c in range(K + 1)]

    def send_schedule(self, r, K):
        """sched[c] = [(q, a, b), ...]: the pieces of runs[r][q] (q != r) rank r sends once it has traced its chunks
        0..c -- the planes no later chunk reaches, not sent before; the last chunk sends everything that is left.  Every
        plane of every run is sent exactly once.  (Memoised: an iteration must not pay for this again.)"""
        K = max(1, int(K))
        cache = self.__dict__.setdefault("_send_cache", {})
        if (r, K) not in cache:
            cache[(r, K)] = self._send_schedule(r, K)
        return cache[(r, K)]

    def _send_schedule(self, r, K):
        n = self.i0[r].size
        b = self.chunk_bounds(n, K)
        sent = np.zeros(self.N, dtype=bool)
        sched = []
        for c in range(K):
            later = self._reach_of(self.i0[r][b[c + 1]:]) if c < K - 1 else np.zeros(self.N, dtype=bool)
            pieces = []
            for q in range(self.P):
                if q == r:
                    continue
                for a0, a1 in self.runs[r][q]:
                    ready = ~later[a0:a1] & ~sent[a0:a1]
                    for s0, s1 in _runs_of(ready):
                        pieces.append((q, a0 + s0, a0 + s1))
                        sent[a0 + s0:a0 + s1] = True
            sched.append(pieces)
        return sched

    def recv_schedule(self, q, K):
        """rsched[c] = [(r, a, b), ...]: what the other ranks send to q after their chunk c (rank order).  Memoised."""
        cache = self.__dict__.setdefault("_recv_cache", {})
        if (q, int(K)) not in cache:
            cache[(q, int(K))] = self._recv_schedule(q, K)
        return cache[(q, int(K))]

    def _recv_schedule(self, q, K):
        out = [[] for _ in range(max(1, int(K)))]
        for r in range(self.P):
            if r == q:
                continue
            for c, pieces in enumerate(self.send_schedule(r, K)):
                out[c] += [(r, a, b) for (dest, a, b) in pieces if dest == q]
        return out

    def work_runs(self, r):
        """Planes rank r zeroes its accumulators on and forms nHI on: what its sources reach plus what it owns.  Memoised."""
        cache = self.__dict__.setdefault("_work_cache", {})
        if r not in cache:
            mask = self.reach[r].copy()
            a, b = self.own[r]
            mask[a:b] = True
            cache[r] = _runs_of(mask)
        return cache[r]

    def reach_runs(self, r):
        return _runs_of(self.reach[r])

    def bytes_per_rank(self, r):
        """(sent, received) by rank r in ONE of the two exchanges of an iteration, in bytes."""
        plane = 8 * self.N * self.N
        sent = sum((b - a) for q in range(self.P) if q != r for a, b in self.runs[r][q])
        recv = sum((b - a) for q in range(self.P) if q != r for a, b in self.runs[q][r])
        return sent * plane, recv * plane

    def largest_transfer(self):
        """Bytes of the largest rank-to-rank transfer of one exchange (what a single xGMI link carries in one direction)."""
        plane = 8 * self.N * self.N
        return plane * max([sum(b - a for a, b in self.runs[r][q]) for r in range(self.P) for q in range(self.P) if q != r] or [0])


class _DevicePointer:
    """Zero-copy view of a library-owned device buffer for torch (__cuda_array_interface__ v3)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {
            "shape": (int(nelem),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
        }


class _Phases:
    """Where one multi-rank iteration spends its time (``TorchComm.phase_timing``; bench.py's ``phases_ms``).

    With RCCL everything an iteration does is ordered on the library's stream, transfers included (the stream waits for
    them, not the host), so a phase is the span between two events recorded on that stream: time the GPU spent on it
    INCLUDING what it waited for -- `wait_rates_add` is the part of the rate exchange the trace did not hide.  The events
    are resolved later (``TorchComm.phase_report``), never inside an iteration.  With gloo (CPU rehearsals) the planes go
    through the host and every phase ends with a host synchronisation and a wall-clock reading."""

    def __init__(self, comm, libasora):
        import time
        self._comm, self._lib, self._clock = comm, libasora, time.perf_counter
        self._device = comm._backend() == "nccl"
        if self._device:
            import torch
            self._torch = torch
            self._stream = comm._library_stream(libasora)
            self._events = [("", self._record())]
        else:
            libasora.synchronize()
            self._t = self._clock()

    def _record(self):
        e = self._torch.cuda.Event(enable_timing=True)
        e.record(self._stream)
        return e

    def mark(self, name):
        """The phase `name` ends here."""
        if self._device:
            self._events.append((name, self._record()))
        else:
            self._lib.synchronize()
            t = self._clock()
            self._comm._phase_add(name, (t - self._t) * 1e3)
            self._t = t

    def close(self):
        self._comm._phase_n += 1
        if self._device:
            self._comm._phase_pending.append(self._events)
            if len(self._comm._phase_pending) >= 512:
                self._comm._phase_resolve()


class TorchComm:
    """Communicator over a torch.distributed process group."""

    def __init__(self, group=None, overlap=None, chunks=8, pipeline_chemistry=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("TorchComm: torch.distributed is not initialised "
                               "(call pyc2ray_amd.dist.init_process_group_from_env() first)")
        self._dist = dist
        self._group = group
        if overlap is None:
            overlap = os.environ.get("PYC2RAY_AMD_OVERLAP", "0") == "1"
        self.overlap = bool(overlap)
        self.chunks = int(os.environ.get("PYC2RAY_AMD_OVERLAP_CHUNKS", chunks))
        # also start each slab's chemistry behind its sum (measured on one GPU: the chemistry kernels then compete
        # with the trace for the memory system and the step gets 0.25 ms LONGER, so this stays off)
        if pipeline_chemistry is None:
            pipeline_chemistry = os.environ.get("PYC2RAY_AMD_OVERLAP_CHEMISTRY", "0") == "1"
        self.pipeline_chemistry = bool(pipeline_chemistry)
        self._comm_stream = None
        #: how the per-rank rate grids are summed: "slab" (SlabPlan: planes to their owners, slab chemistry, xh_av
        #: back) or "allreduce" (full-grid all-reduce, chemistry replicated on every rank)
        self.exchange = os.environ.get("PYC2RAY_AMD_EXCHANGE", "slab")
        #: "allreduce" without `overlap`: True = on the device-resident loop (``reduce_begin``: batches of iterations per host round
        #: trip, as the slab exchange and the one-GPU loop); False = three calls and a host read-back per iteration
        self.device_loop = os.environ.get("PYC2RAY_AMD_REDUCE_LOOP", "1") != "0"
        #: slab exchange: trace chunks per iteration; planes that are final after a chunk travel while the next is traced
        # (measured per-rank compute + modelled links, tools/slab_compute_model.py, profiles/r03_slab_model_chunks.txt: splitting
        #  the trace costs more than the early sends hide from four ranks on -- 125 sources per launch no longer fill the chip
        #  -- and gains ~15 % with two ranks)
        self.slab_chunks = int(os.environ.get("PYC2RAY_AMD_SLAB_CHUNKS", "2" if self._dist.get_world_size(group) == 2 else "1"))
        #: every rank derives the convergence decision from the SAME all-reduced scalars (no broadcast of the decision needed)
        self.identical_scalars = True
        #: True: every iteration books where its time went (``_Phases``); read with ``phase_report``
        self.phase_timing = False
        self._phase_ms, self._phase_n, self._phase_pending = {}, 0, []
        # Bring the communicator up with a collective EVERY rank takes part in.  The slab exchange is point-to-point
        # and a rank with nothing to send or receive skips it; if that were the first operation on the process group,
        # the ranks that do take part would wait for the others in the communicator's set-up.
        import torch
        t = torch.zeros(1, dtype=torch.float64)
        if self._backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, group=self._group)

    # -- mpi4py-flavoured surface ---------------------------------------------------------------
    def Get_rank(self):
        return self._dist.get_rank(self._group)

    def Get_size(self):
        return self._dist.get_world_size(self._group)

    def Barrier(self):
        self._dist.barrier(self._group)

    @staticmethod
    def _buf(spec):
        return spec[0] if isinstance(spec, (list, tuple)) else spec

    def _backend(self):
        return self._dist.get_backend(self._group)

    @property
    def backend(self):
        """"nccl" (RCCL: exchanges on the device, batches of iterations per host round trip) or "gloo" (through the host)."""
        return self._backend()

    def _tensor_of(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.cuda() if self._backend() == "nccl" else t

    def Allreduce(self, sendbuf, recvbuf, op=None):
        """Sum-allreduce of a numpy buffer (mpi4py calling convention, IN_PLACE supported)."""
        out = self._buf(recvbuf)
        in_place = isinstance(sendbuf, str) and sendbuf == MPI.IN_PLACE
        src = out if in_place else self._buf(sendbuf)
        t = self._tensor_of(src)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        out[...] = t.cpu().numpy().reshape(out.shape)

    def Reduce(self, sendbuf, recvbuf, op=None, root=0):
        """Sum-reduce to `root` (evolve.py:433-436 convention: root passes IN_PLACE + its buffer)."""
        in_place = isinstance(sendbuf, str) and sendbuf == MPI.IN_PLACE
        src = self._buf(recvbuf) if in_place else self._buf(sendbuf)
        t = self._tensor_of(src)
        self._dist.reduce(t, dst=root, op=self._dist.ReduceOp.SUM, group=self._group)
        if self.Get_rank() == root:
            out = self._buf(recvbuf)
            out[...] = t.cpu().numpy().reshape(out.shape)

    def Bcast(self, buf, root=0):
        arr = self._buf(buf)
        a = np.asarray(arr)
        t = self._tensor_of(a)
        self._dist.broadcast(t, src=root, group=self._group)
        res = t.cpu().numpy().reshape(a.shape)
        if isinstance(arr, np.ndarray):
            arr[...] = res
        else:                       # array.array and friends (evolve.py:484-487)
            for i, v in enumerate(res.ravel()):
                arr[i] = v.item()

    # -- the data-path collective -------------------------------------------------------------------
    def allreduce_device_grid(self, libasora, which, N):
        """In-place sum over ranks of the device-resident grid `which` (N^3 float64)."""
        import torch
        if self.Get_size() == 1 and os.environ.get("PYC2RAY_AMD_FORCE_COLLECTIVE", "0") != "1":
            return          # nothing to sum (the env switch lets a 1-GPU box exercise the collective path)
        if self._backend() == "nccl":
            libasora.synchronize()                     # the library works on its own stream
            view = torch.as_tensor(_DevicePointer(libasora.device_ptr(which), N ** 3), device="cuda")
            self._dist.all_reduce(view, op=self._dist.ReduceOp.SUM, group=self._group)
            torch.cuda.synchronize()
        else:
            host = libasora.grid_to_host(which, np.empty((N, N, N)))
            t = torch.from_numpy(host)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            libasora.grid_to_device(which, host)


    # -- diagnostics: per-phase times of an iteration, measured link rates ----------------------------------------
    def _phase_add(self, name, ms):
        self._phase_ms[name] = self._phase_ms.get(name, 0.0) + ms

    def _phase_resolve(self):
        if not self._phase_pending:
            return
        import torch
        torch.cuda.synchronize()
        for events in self._phase_pending:
            for (_, e0), (name, e1) in zip(events[:-1], events[1:]):
                self._phase_add(name, e0.elapsed_time(e1))
        self._phase_pending = []

    def phase_reset(self):
        self._phase_resolve()
        self._phase_ms, self._phase_n = {}, 0

    def phase_report(self, reduce_max=True):
        """Mean milliseconds per iteration of every phase booked since ``phase_reset`` -- the MAXIMUM over the ranks when
        `reduce_max` (a collective: every rank must call it), plus "iterations".  Phase names: slab exchange --
        trace_fold_post, wait_rates_add, slab_pass, xh_av_exchange_nhi, scalar_allreduce_test; all-reduce loop -- trace_fold,
        rate_allreduce, pass_test; pipelined all-reduce (raytrace_and_allreduce) -- trace, rate_allreduce, chemistry."""
        import torch
        self._phase_resolve()
        names = sorted(self._phase_ms)
        n = max(self._phase_n, 1)
        vals = [self._phase_ms[k] / n for k in names]
        if reduce_max and self.Get_size() > 1:
            # (ranks on different exchange paths would book different names: the caller keeps them on one path)
            t = torch.tensor(vals, dtype=torch.float64)
            if self._backend() == "nccl":
                t = t.cuda()
            self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
            vals = t.cpu().tolist()
        out = dict(zip(names, vals))
        out["iterations"] = self._phase_n
        return out

    def measure_links(self, p2p_bytes=16 << 20, allreduce_bytes=128 << 20, reps=3, p2p=True):
        """What the links of THIS job deliver, through the calls the data path uses: a ring of point-to-point transfers of
        `p2p_bytes` (every rank sends to its right neighbour and receives from its left one with ``batch_isend_irecv``, as the
        slab exchange does) and an in-place sum all-reduce of `allreduce_bytes` (the full-grid exchange).  One untimed round
        each, then the fastest of `reps`; times are the MAXIMUM over the ranks, so every rank returns the same numbers and
        derives the same choice from them.  p2p = False leaves the ring out (a caller whose point-to-point preflight failed:
        the all-reduce time is still worth having).  Returns {"p2p_bytes", "p2p_ms", "p2p_GBs" (per link and direction),
        "allreduce_bytes", "allreduce_ms", "allreduce_busbw_GBs" (2 (P-1)/P bytes / time, the ring's per-link rate)}."""
        import time
        import torch
        dist = self._dist
        P, me = self.Get_size(), self.Get_rank()
        nccl = self._backend() == "nccl"
        dev = "cuda" if nccl else "cpu"

        def fence():
            if nccl:
                torch.cuda.synchronize()

        def timed(fn):
            best = None
            for rep in range(reps + 1):
                dist.barrier(self._group)
                fence()
                t0 = time.perf_counter()
                fn()
                fence()
                dt = time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            t = torch.tensor([best], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._group)
            return float(t.item())

        out = {"p2p_bytes": int(p2p_bytes), "allreduce_bytes": int(allreduce_bytes), "ranks": P, "backend": self._backend()}
        if P < 2:
            out.update(p2p_ms=None, p2p_GBs=None, allreduce_ms=None, allreduce_busbw_GBs=None)
            return out
        n1 = max(1, int(p2p_bytes) // 8)
        snd = torch.full((n1,), float(me + 1), dtype=torch.float64, device=dev)
        rcv = torch.zeros((n1,), dtype=torch.float64, device=dev)

        def ring():
            ops = [dist.P2POp(dist.isend, snd, (me + 1) % P, group=self._group),
                   dist.P2POp(dist.irecv, rcv, (me - 1) % P, group=self._group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        t_p2p = timed(ring) if p2p else None
        n2 = max(1, int(allreduce_bytes) // 8)
        buf = torch.zeros((n2,), dtype=torch.float64, device=dev)
        t_ar = timed(lambda: dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self._group))
        del buf, snd, rcv
        out.update(p2p_ms=t_p2p * 1e3 if p2p else None, p2p_GBs=8.0 * n1 / t_p2p / 1e9 if p2p else None,
                   allreduce_ms=t_ar * 1e3, allreduce_busbw_GBs=2.0 * (P - 1) / P * 8.0 * n2 / t_ar / 1e9)
        return out

    def preflight_p2p(self, nelem=65536):
        """One small point-to-point round (every rank sends to its right neighbour and receives from its left one) through
        the same call the slab exchange uses.  The exchange has never run between real GPUs on the build box; a caller
        that gets an exception here (or False from any rank, after a MIN-reduce of the results) takes the all-reduce path.
        Returns True when the round went through on this rank and the payload arrived intact."""
        import torch
        dist = self._dist
        P, me = self.Get_size(), self.Get_rank()
        if P < 2:
            return True
        dev = "cuda" if self._backend() == "nccl" else "cpu"
        out = torch.full((nelem,), float(me + 1), dtype=torch.float64, device=dev)
        inc = torch.zeros((nelem,), dtype=torch.float64, device=dev)
        ops = [dist.P2POp(dist.isend, out, (me + 1) % P, group=self._group),
               dist.P2POp(dist.irecv, inc, (me - 1) % P, group=self._group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if dev == "cuda":
            torch.cuda.synchronize()
        return bool((inc == float((me - 1) % P + 1)).all().item())

    # -- slab exchange: rates to the owners of the planes, ionised fraction back (see SlabPlan) -------------------
    @staticmethod
    def shard_sources_by_slab(src_pos, src_flux, nprocs):
        """Order the source list by its first coordinate and cut it into the reference's contiguous blocks
        (pyc2ray/evolve.py:362-367: perrank = NumSrc // nprocs, the last rank takes the remainder).  Returns
        (src_pos, src_flux, bounds) with rank r holding [bounds[r], bounds[r+1]) of the reordered list."""
        pos, flux = np.asarray(src_pos), np.asarray(src_flux)
        order = np.argsort(pos[0], kind="stable")
        n = flux.shape[0]
        per = n // nprocs
        bounds = [r * per for r in range(nprocs)] + [n]
        return pos[:, order], flux[order], bounds

    def _planes_view(self, libasora, which, N):
        """Zero-copy (N, N*N) view of a library grid; kept per (address, N): the grids do not move between iterations."""
        import torch
        ptr = libasora.device_ptr(which)
        cache = self.__dict__.setdefault("_views", {})
        if (ptr, N) not in cache:
            cache[(ptr, N)] = torch.as_tensor(_DevicePointer(ptr, N ** 3), device="cuda").view(N, N * N)
        return cache[(ptr, N)]

    def _library_stream(self, libasora):
        import torch
        ptr = libasora.stream_ptr()
        if getattr(self, "_lib_stream_ptr", None) != ptr:
            self._lib_stream, self._lib_stream_ptr = torch.cuda.ExternalStream(ptr), ptr
        return self._lib_stream

    def _post(self, libasora, which, N, sends, recvs, add, tag):
        """Start one round of plane transfers of grid `which`: `sends` = [(peer, a, b)] planes [a, b) to peer, `recvs` =
        [(peer, a, b)] planes from peer.  Returns a handle for ``_complete``.  With RCCL everything is ordered on the
        library's stream: the sends start behind the kernels that produced the planes, and nothing waits on the host; the
        transfers themselves run on RCCL's stream, beside whatever the library's stream does next.  add = True: what is
        received is ADDED to the planes (in ``_complete``, in the order of `recvs`), else it replaces them."""
        import torch
        dist = self._dist
        if not sends and not recvs:
            return None
        if self._backend() == "nccl":
            lib_stream = self._library_stream(libasora)
            with torch.cuda.stream(lib_stream):
                grid = self._planes_view(libasora, which, N)
                # staging buffers, receive targets and the P2POp list are the same every iteration: built once per round
                key = (tag, which, N, add, tuple(sends), tuple(recvs), grid.data_ptr())
                cache = self.__dict__.setdefault("_rounds", {})
                if key not in cache:
                    if add:
                        targets = [torch.empty((b - a, N * N), dtype=torch.float64, device="cuda") for _, a, b in recvs]
                    else:
                        targets = [grid[a:b] for _, a, b in recvs]
                    ops = [dist.P2POp(dist.isend, grid[a:b], q, group=self._group) for q, a, b in sends]
                    ops += [dist.P2POp(dist.irecv, t, q, group=self._group) for (q, _, _), t in zip(recvs, targets)]
                    cache[key] = (targets, ops)
                targets, ops = cache[key]
                works = dist.batch_isend_irecv(ops)
            return ("nccl", works, recvs, targets, add)
        # gloo (CPU tests, several ranks on one GPU): staged through the host
        out = [torch.from_numpy(libasora.planes_to_host(which, a, b - a, N)) for _, a, b in sends]
        inc = [torch.empty((b - a, N, N), dtype=torch.float64) for _, a, b in recvs]
        ops = [dist.P2POp(dist.isend, t, q, group=self._group) for (q, _, _), t in zip(sends, out)]
        ops += [dist.P2POp(dist.irecv, t, q, group=self._group) for (q, _, _), t in zip(recvs, inc)]
        works = dist.batch_isend_irecv(ops)
        return ("gloo", works, recvs, inc, add, out)

    def _complete(self, libasora, which, N, handle):
        """Wait for a round started by ``_post`` and put what was received in place (fixed order: same bits on every run)."""
        import torch
        if handle is None:
            return
        kind, works, recvs, targets, add = handle[:5]
        if kind == "nccl":
            lib_stream = self._library_stream(libasora)
            with torch.cuda.stream(lib_stream):
                for w in works:
                    w.wait()                       # the library's stream waits, not the host
                if add:
                    grid = self._planes_view(libasora, which, N)
                    for (q, a, b), t in zip(recvs, targets):
                        grid[a:b] += t
            return
        for w in works:
            w.wait()
        for (q, a, b), t in zip(recvs, targets):
            if add:
                mine = libasora.planes_to_host(which, a, b - a, N)
                libasora.planes_to_device(which, a, mine + t.numpy())
            else:
                libasora.planes_to_device(which, a, t.numpy())

    # -- the device-resident loop over several ranks (asora_evolve_slab_*, include/asora_hip.h) ------------------------
    def slab_begin(self, libasora, plan, N, R, sig, dr, num_src_local, minlogtau, dlogtau, NumTau, chemistry,
                   conv_criterion, convergence_fraction):
        """Start a time step of the sharded loop: NDENS, TEMP, XH and this rank's sources are on the device.  `chemistry` =
        (dt, bh00, albpow, colh0, temph0, abu_c); conv_criterion from the TOTAL source count (pyc2ray/evolve.py:127,346)."""
        a, b = plan.own[self.Get_rank()]
        libasora.evolve_begin_slab(*chemistry, R, sig, dr, minlogtau, dlogtau, NumTau, 0, num_src_local, conv_criterion,
                                   convergence_fraction, a, b - a)
        self._slab = (plan, int(N), int(num_src_local))

    def reduce_begin(self, libasora, N, R, sig, dr, num_src_local, minlogtau, dlogtau, NumTau, chemistry, conv_criterion,
                     convergence_fraction):
        """Start a time step of the same device-resident loop with the reference's exchange (pyc2ray/evolve.py:433-437): every
        rank traces its sources, the rate grid is all-reduced, every rank runs the chemistry of the WHOLE grid on identical
        rates (and so takes the same decisions without exchanging anything else).  ``slab_enqueue`` / ``slab_poll`` drive it."""
        libasora.evolve_begin_slab(*chemistry, R, sig, dr, minlogtau, dlogtau, NumTau, 0, num_src_local, conv_criterion,
                                   convergence_fraction, 0, N)
        self._slab = (None, int(N), int(num_src_local))

    def _reduce_one(self, libasora):
        """One iteration of the loop begun with ``reduce_begin``: trace -> both accumulator layouts of all planes folded into
        the out-box -> the out-box summed over the ranks in place -> the fused pass on the whole grid reading the out-box (it
        keeps the summed rates in PHI_ION) -> convergence test on the pass's own sums.  With RCCL the all-reduce is ordered on
        the library's stream and nothing waits on the host; with gloo the out-box goes through the host."""
        import torch
        _, N, num_src_local = self._slab
        ph = _Phases(self, libasora) if self.phase_timing else None
        libasora.evolve_slab_trace(0, num_src_local)
        libasora.evolve_slab_fold_all()
        if ph: ph.mark("trace_fold")
        if self.Get_size() > 1 or os.environ.get("PYC2RAY_AMD_FORCE_COLLECTIVE", "0") == "1":
            if self._backend() == "nccl":
                with torch.cuda.stream(self._library_stream(libasora)):
                    self._dist.all_reduce(self._outbox_view(libasora, N), op=self._dist.ReduceOp.SUM, group=self._group)
            else:
                host = libasora.evolve_slab_outbox_to_host(0, N, N)
                self._dist.all_reduce(torch.from_numpy(host), op=self._dist.ReduceOp.SUM, group=self._group)
                libasora.evolve_slab_outbox_from_host(0, host)
        if ph: ph.mark("rate_allreduce")
        libasora.evolve_slab_pass()
        libasora.evolve_slab_close(None)             # (the pass's sums are those of the whole grid, the same on every rank)
        if ph:
            ph.mark("pass_test")
            ph.close()

    def slab_enqueue(self, libasora, iterations=1):
        """Enqueue outer iterations of the step begun with ``slab_begin``.  One iteration = what the one-GPU loop does -- the
        trace and ONE fused pass -- with two plane exchanges in between:

          trace (in K chunks; after each, the foreign planes no later chunk reaches are folded into the out-box and sent to
          their owners) -> received rates added on the own planes -> fused pass on the own planes (rates folded, chemistry,
          nHI in both layouts, next accumulators zeroed) -> new xh_av of the planes other ranks trace through sent back, nHI
          formed on the halo planes as they arrive -> the three convergence sums all-reduced IN PLACE on the device and the
          test of evolve.py:216-236 evaluated there.

        With RCCL nothing here waits on the host: transfers, kernels and the test are ordered on the library's stream, every
        launch is gated by the device's `done` flag, and every rank evaluates the test on identical bits, so all ranks stop
        at the same iteration; ``slab_poll`` reads the status back once per batch.  With gloo (CPU rehearsals) the planes
        and the sums are staged through the host, one synchronisation per exchange."""
        one = self._reduce_one if self._slab[0] is None else self._slab_one
        for _ in range(int(iterations)):
            one(libasora)

    def slab_poll(self, libasora, max_rows=32):
        """(iterations carried out, converged, rows) -- asora_evolve_poll; folds the last iteration's rates into PHI_ION
        (complete on the own planes; ``slab_gather`` collects the owners' slabs at the end of the step)."""
        return libasora.evolve_poll(max_rows)

    def _outbox_view(self, libasora, N):
        import torch
        ptr = libasora.evolve_slab_outbox_ptr()
        cache = self.__dict__.setdefault("_views", {})
        if (ptr, N) not in cache:
            cache[(ptr, N)] = torch.as_tensor(_DevicePointer(ptr, N ** 3), device="cuda").view(N, N * N)
        return cache[(ptr, N)]

    def _post_rates(self, libasora, N, sends, recvs, tag):
        """The first exchange: out-box planes to their owners, foreign contributions to the own planes into receive buffers."""
        import torch
        dist = self._dist
        if not sends and not recvs:
            return None
        if self._backend() == "nccl":
            with torch.cuda.stream(self._library_stream(libasora)):
                box = self._outbox_view(libasora, N)
                key = ("rates", tag, N, tuple(sends), tuple(recvs), box.data_ptr())
                cache = self.__dict__.setdefault("_rounds", {})
                if key not in cache:
                    targets = [torch.empty((b - a, N * N), dtype=torch.float64, device="cuda") for _, a, b in recvs]
                    ops = [dist.P2POp(dist.isend, box[a:b], q, group=self._group) for q, a, b in sends]
                    ops += [dist.P2POp(dist.irecv, t, q, group=self._group) for (q, _, _), t in zip(recvs, targets)]
                    cache[key] = (targets, ops)
                targets, ops = cache[key]
                works = dist.batch_isend_irecv(ops)
            return ("nccl", works, recvs, targets)
        out = [torch.from_numpy(libasora.evolve_slab_outbox_to_host(a, b - a, N)) for _, a, b in sends]
        inc = [torch.empty((b - a, N, N), dtype=torch.float64) for _, a, b in recvs]
        ops = [dist.P2POp(dist.isend, t, q, group=self._group) for (q, _, _), t in zip(sends, out)]
        ops += [dist.P2POp(dist.irecv, t, q, group=self._group) for (q, _, _), t in zip(recvs, inc)]
        return ("gloo", dist.batch_isend_irecv(ops), recvs, inc, out)

    def _complete_rates(self, libasora, handle):
        """Wait for a round of ``_post_rates`` and add what arrived, in the order of `recvs` (rank order)."""
        import torch
        if handle is None:
            return
        kind, works, recvs, targets = handle[:4]
        if kind == "nccl":
            with torch.cuda.stream(self._library_stream(libasora)):
                for w in works:
                    w.wait()                                   # the library's stream waits, not the host
            for (_, a, b), t in zip(recvs, targets):
                libasora.evolve_slab_add(a, b - a, t.data_ptr())
            return
        for w in works:
            w.wait()
        for (_, a, b), t in zip(recvs, targets):
            libasora.evolve_slab_add_host(a, t.numpy())

    def _close_iteration(self, libasora):
        """The three sums of this rank's pass summed over the ranks, then the convergence test on the device."""
        import torch
        if self._backend() == "nccl" and hasattr(libasora, "reduction_ptr"):
            ptr = libasora.reduction_ptr()
            cache = self.__dict__.setdefault("_views", {})
            if (ptr, 3) not in cache:
                cache[(ptr, 3)] = torch.as_tensor(_DevicePointer(ptr, 3), device="cuda")
            with torch.cuda.stream(self._library_stream(libasora)):
                self._dist.all_reduce(cache[(ptr, 3)], op=self._dist.ReduceOp.SUM, group=self._group)
            libasora.evolve_slab_close(None)
            return
        part = libasora.chemistry_finish()                      # (conv_flag, sum x, sum 1-x) of this rank
        t = torch.tensor([float(part[0]), float(part[1]), float(part[2])], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        v = t.tolist()
        libasora.evolve_slab_close((v[0], v[1], v[2]))

    def _slab_one(self, libasora):
        from . import _capi
        plan, N, num_src_local = self._slab
        me = self.Get_rank()
        # every rank walks through the same number of rounds and derives every other rank's schedule: the chunk count is a
        # function of the plan, never of this rank alone
        K = plan.common_chunks(getattr(self, "slab_chunks", 1))
        ph = _Phases(self, libasora) if self.phase_timing else None
        sched, rsched = plan.send_schedule(me, K), plan.recv_schedule(me, K)
        bounds = plan.chunk_bounds(num_src_local, K)
        handles = []
        for c in range(K):
            libasora.evolve_slab_trace(bounds[c], bounds[c + 1] - bounds[c])
            for _, a, b in sched[c]:
                libasora.evolve_slab_fold_out(a, b - a)                 # the planes that leave now
            handles.append(self._post_rates(libasora, N, sched[c], rsched[c], c))
        if ph: ph.mark("trace_fold_post")
        for h in handles:                                               # chunk order, then rank order: a fixed order of additions
            self._complete_rates(libasora, h)
        if ph: ph.mark("wait_rates_add")
        libasora.evolve_slab_pass()
        if ph: ph.mark("slab_pass")
        # xh_av back: the owner q of a run sends it to the rank r that traces through it; nHI there once it has arrived
        back = plan.__dict__.setdefault("_back_cache", {})
        if me not in back:
            back[me] = ([(r, s0, s1) for r in range(plan.P) if r != me for s0, s1 in plan.runs[r][me]],
                        [(q, s0, s1) for q in range(plan.P) if q != me for s0, s1 in plan.runs[me][q]])
        sends, recvs = back[me]
        self._complete(libasora, _capi.GRID_XH_AV, N, self._post(libasora, _capi.GRID_XH_AV, N, sends, recvs, False, "xh_av"))
        for _, a, b in recvs:
            libasora.evolve_slab_nhi(a, b - a)
        if ph: ph.mark("xh_av_exchange_nhi")
        self._close_iteration(libasora)
        if ph:
            ph.mark("scalar_allreduce_test")
            ph.close()

    def slab_iteration(self, libasora, plan, N, R, sig, dr, num_src_local, minlogtau, dlogtau, NumTau, chemistry, first):
        """ONE outer iteration and its three sums (conv_flag, sum x, sum 1-x) over the WHOLE grid, identical on every rank:
        ``slab_begin`` (when `first`; with a convergence test that never passes) + ``slab_enqueue(1)`` + ``slab_poll``.  For
        callers that drive the loop themselves; evolve3D_MPI and bench.py enqueue batches instead."""
        if first:
            self.slab_begin(libasora, plan, N, R, sig, dr, num_src_local, minlogtau, dlogtau, NumTau, chemistry, -1.0, 0.0)
        self.slab_enqueue(libasora, 1)
        _, _, rows = self.slab_poll(libasora, 1)
        return int(rows[-1][0]), float(rows[-1][1]), float(rows[-1][2])

    def slab_gather(self, libasora, plan, which, N):
        """Every rank gets every owner's slab of grid `which` (end of a time step: xh_intermed, phi_ion)."""
        import torch
        me = self.Get_rank()
        if self._backend() == "nccl":
            lib_stream = torch.cuda.ExternalStream(libasora.stream_ptr())
            with torch.cuda.stream(lib_stream):
                grid = self._planes_view(libasora, which, N)
                for q, (a, b) in enumerate(plan.own):
                    if b > a:
                        self._dist.broadcast(grid[a:b], src=q, group=self._group)
            return
        for q, (a, b) in enumerate(plan.own):
            if b <= a:
                continue
            t = torch.from_numpy(libasora.planes_to_host(which, a, b - a, N)) if q == me else \
                torch.empty((b - a, N, N), dtype=torch.float64)
            self._dist.broadcast(t, src=q, group=self._group)
            if q != me:
                libasora.planes_to_device(which, a, t.numpy())

    # -- raytrace + sum over ranks, optionally pipelined ------------------------------------------------
    @staticmethod
    def sort_sources_for_overlap(src_pos, src_flux):
        """Order a rank's sources by their first coordinate (what the pipelined path needs).  src_pos is
        (3, n) 1-based; returns (src_pos, src_flux) reordered.  The sum over sources is order-independent
        up to floating-point rounding."""
        order = np.argsort(np.asarray(src_pos)[0], kind="stable")
        return np.asarray(src_pos)[:, order], np.asarray(src_flux)[order]

    @staticmethod
    def final_plane_runs(N, chunks, R, c, reduced):
        """Planes of the rate grid that are final once the sources of chunks 0..c (sources with 0-based first
        coordinate < (c+1)*N//chunks) have been traced and that have not been summed yet, as a list of
        half-open runs.  A later source sits at i >= (c+1)*N//chunks and reaches the planes i-R..i+R, wrapping
        at most onto planes < R; the last chunk releases everything that is left.  Depends on (N, chunks, R)
        only, so every rank derives the same runs."""
        margin = int(np.ceil(R)) + 1
        if c == chunks - 1:
            want = ~reduced
        else:
            want = np.zeros(N, dtype=bool)
            lo, hi = margin, (c + 1) * N // chunks - margin
            if hi > lo:
                want[lo:hi] = True
            want &= ~reduced
        runs, i = [], 0
        while i < N:
            if want[i]:
                j = i
                while j < N and want[j]:
                    j += 1
                runs.append((i, j))
                i = j
            else:
                i += 1
        return runs

    def raytrace_and_allreduce(self, libasora, N, R, sig, dr, num_src_local, minlogtau, dlogtau, NumTau,
                               src_i0=None, chemistry=None):
        """Trace this rank's sources and leave the sum over ranks of the rate grids in the device-resident
        phi_ion grid.  Pipelined when self.overlap is set and `src_i0` (0-based first coordinates of the local
        sources, ascending, in upload order) is given; otherwise trace, then all-reduce.

        With `chemistry` = (dt, bh00, albpow, colh0, temph0, abu_c) the chemistry pass is run as well and its
        (conv_flag, sum x, sum 1-x) returned; with self.pipeline_chemistry each slab's chemistry starts as soon as its
        rates are summed, under the remaining trace and all-reduces."""
        from . import _capi
        if not self.overlap or src_i0 is None:
            if not self.phase_timing:
                libasora.raytrace_device(R, sig, dr, 0, num_src_local, minlogtau, dlogtau, NumTau)
                self.allreduce_device_grid(libasora, _capi.GRID_PHI_ION, N)
                return libasora.chemistry_device(*chemistry) if chemistry is not None else None
            # the same three calls between wall-clock readings (each of them ends with a host synchronisation anyway)
            import time
            t0 = time.perf_counter()
            libasora.raytrace_device(R, sig, dr, 0, num_src_local, minlogtau, dlogtau, NumTau)
            libasora.synchronize()
            t1 = time.perf_counter()
            self.allreduce_device_grid(libasora, _capi.GRID_PHI_ION, N)
            t2 = time.perf_counter()
            res = libasora.chemistry_device(*chemistry) if chemistry is not None else None
            t3 = time.perf_counter()
            self._phase_add("trace", (t1 - t0) * 1e3)
            self._phase_add("rate_allreduce", (t2 - t1) * 1e3)
            self._phase_add("chemistry", (t3 - t2) * 1e3)
            self._phase_n += 1
            return res
        import torch
        src_i0 = np.asarray(src_i0)
        if src_i0.size and np.any(np.diff(src_i0) < 0):
            raise ValueError("raytrace_and_allreduce: sources must be uploaded in ascending order of their first "
                             "coordinate (TorchComm.sort_sources_for_overlap)")
        K = max(1, min(self.chunks, N))
        starts = np.searchsorted(src_i0, [c * N // K for c in range(K + 1)], side="left")
        starts[-1] = src_i0.size
        nccl = self._backend() == "nccl"
        force = os.environ.get("PYC2RAY_AMD_FORCE_COLLECTIVE", "0") == "1"
        collective = self.Get_size() > 1 or force
        if nccl and collective:
            lib_stream = torch.cuda.ExternalStream(libasora.stream_ptr())
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            view = torch.as_tensor(_DevicePointer(libasora.device_ptr(_capi.GRID_PHI_ION), N ** 3), device="cuda")
        reduced = np.zeros(N, dtype=bool)
        first_slab = True
        slab_chemistry = chemistry is not None and self.pipeline_chemistry
        libasora.raytrace_begin(R, sig, dr, minlogtau, dlogtau, NumTau)
        for c in range(K):
            libasora.raytrace_range(int(starts[c]), int(starts[c + 1] - starts[c]))
            for a, b in self.final_plane_runs(N, K, R, c, reduced):
                libasora.raytrace_fold(a, b - a)
                reduced[a:b] = True
                if collective and nccl:
                    done = torch.cuda.Event()
                    done.record(lib_stream)                    # the slab is final once the library stream gets here
                    self._comm_stream.wait_event(done)
                    with torch.cuda.stream(self._comm_stream):
                        self._dist.all_reduce(view[a * N * N:b * N * N], op=self._dist.ReduceOp.SUM, group=self._group)
                    if slab_chemistry:                         # the slab's chemistry goes behind its sum
                        summed = torch.cuda.Event()
                        summed.record(self._comm_stream)
                        lib_stream.wait_event(summed)
                elif collective:                               # gloo (CPU tests): staged through the host
                    host = libasora.grid_to_host(_capi.GRID_PHI_ION, np.empty((N, N, N)))
                    t = torch.from_numpy(host[a:b])
                    self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
                    libasora.grid_to_device(_capi.GRID_PHI_ION, host)
                if slab_chemistry:
                    libasora.chemistry_range(*chemistry, a, b - a, first_slab)
                    first_slab = False
        assert reduced.all()
        if nccl and collective:
            lib_stream.wait_stream(self._comm_stream)         # whatever follows on the library stream needs the sums
        if chemistry is None:
            return None
        return libasora.chemistry_finish() if slab_chemistry else libasora.chemistry_device(*chemistry)
