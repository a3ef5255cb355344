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

__all__ = ["MPI", "TorchComm", "init_process_group_from_env"]


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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class _DevicePointer:
    """Zero-copy view of a library-owned device buffer for torch (__cuda_array_interface__ v3)."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {
            "shape": (int(nelem),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
        }


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
            libasora.raytrace_device(R, sig, dr, 0, num_src_local, minlogtau, dlogtau, NumTau)
            self.allreduce_device_grid(libasora, _capi.GRID_PHI_ION, N)
            return libasora.chemistry_device(*chemistry) if chemistry is not None else None
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
