"""Multi-GPU communicator: one process per MI355X under torch.distributed.

The reference shards sources over MPI ranks with mpi4py and moves N^3 float64 grids through host
memory (pyc2ray/evolve.py:433-437,480-497).  The path has exactly one real exchange step per outer
iteration -- the sum of the per-rank photo-ionisation rate grids -- so that is the only collective
here: an all-reduce (backend "nccl" == RCCL over xGMI) applied IN PLACE to the device-resident
phi_ion grid.  With the "gloo" backend (CPU tests) the grid is staged through the host.

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

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("TorchComm: torch.distributed is not initialised "
                               "(call pyc2ray_amd.dist.init_process_group_from_env() first)")
        self._dist = dist
        self._group = group

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
        src = out if sendbuf is MPI.IN_PLACE or sendbuf == MPI.IN_PLACE else self._buf(sendbuf)
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
