"""Result grids in page-locked host memory.

evolve3D returns two fresh N^3 arrays per time step, as the reference does (pyc2ray/evolve.py:244-245).  A download into
a fresh ``np.empty`` array pays a page fault per 4 KiB inside the copy (measured at 256^3 on the MI355X box: 14.7 ms per
grid, 9 GB/s, against 2.4 ms into page-locked memory), and the array the caller drops a step later is unmapped again.
The arrays handed out here are ordinary ``numpy.ndarray`` objects over buffers from ``asora_host_alloc`` (hipHostMalloc);
when the last reference to such an array -- and to every view of it -- is gone, its buffer returns to a small free list and
backs the result of a later step.  Nothing is ever handed out twice while someone can still see it.

When page-locked memory cannot be had (no device, the limits below, an allocation failure) :func:`empty` is
``numpy.empty``: this is about where the bytes land, not about who computes them.
"""
import ctypes
import os
import threading

import numpy as np

#: free buffers kept per size, and the cap on page-locked bytes handed out or kept at any time (a caller that stores every
#: step's output keeps getting arrays, pageable ones, beyond it)
MAX_FREE_PER_SIZE = 4
MAX_PINNED_BYTES = int(float(os.environ.get("PYC2RAY_AMD_PINNED_GIB", "8")) * 2 ** 30)

# re-entrant: _Owner.__del__ may run (cyclic GC) while this thread is inside _take / _give_back and holds the lock
_lock = threading.RLock()
_free = {}              # nbytes -> [address, ...]
_pinned_bytes = 0
_enabled = os.environ.get("PYC2RAY_AMD_PINNED_RESULTS", "1") != "0"


class _Owner:
    """Keeps one page-locked buffer alive for as long as an array (or a view of one) refers to it."""
    __slots__ = ("address", "nbytes", "_lib", "__array_interface__")

    def __init__(self, lib, address, nbytes):
        self._lib, self.address, self.nbytes = lib, address, nbytes
        self.__array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (address, False), "version": 3}

    def __del__(self):
        _give_back(self._lib, self.address, self.nbytes)


def _give_back(lib, address, nbytes):
    global _pinned_bytes
    with _lock:
        held = _free.setdefault(nbytes, [])
        if len(held) < MAX_FREE_PER_SIZE:
            held.append(address)
            return
        _pinned_bytes -= nbytes
    try:
        lib.asora_host_free(ctypes.c_void_p(address))
    except Exception:               # interpreter shutdown: the process's memory goes with it
        pass


def _take(lib, nbytes):
    global _pinned_bytes
    with _lock:
        held = _free.get(nbytes)
        if held:
            return held.pop()
        if _pinned_bytes + nbytes > MAX_PINNED_BYTES:
            return None
        _pinned_bytes += nbytes
    p = ctypes.c_void_p()
    if lib.asora_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(p)) != 0 or not p.value:
        with _lock:
            _pinned_bytes -= nbytes
        return None
    return p.value


def empty(lib, shape, order="C"):
    """``numpy.empty(shape, float64, order)`` over page-locked memory when that is available.  ``lib`` is the loaded
    ctypes library (``load_asora()._lib``)."""
    n = int(np.prod(shape))
    if not _enabled or n == 0:
        return np.empty(shape, order=order)
    address = _take(lib, n * 8)
    if address is None:
        return np.empty(shape, order=order)
    flat = np.asarray(_Owner(lib, address, n * 8))         # flat.base is the owner
    return flat.reshape(shape, order=order)


def stats():
    with _lock:
        return {"pinned_bytes": _pinned_bytes, "free": {k: len(v) for k, v in _free.items()}}


def release_free_buffers(lib):
    """Return the free list to the system (device_close calls this)."""
    global _pinned_bytes
    with _lock:
        drop = [(a, n) for n, held in _free.items() for a in held]
        _free.clear()
        _pinned_bytes -= sum(n for _, n in drop)
    for a, _ in drop:
        lib.asora_host_free(ctypes.c_void_p(a))
