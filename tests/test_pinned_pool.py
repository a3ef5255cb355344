"""The pool behind the arrays evolve3D returns (pyc2ray_amd/_pinned.py): a buffer goes back to the free list only when
the array AND every view of it are gone, is then reused, and the caps fall back to ordinary numpy memory.  The allocator
here is libc's malloc behind the two C-ABI names, so this runs without a GPU; tests/test_gpu_parity.py checks the real one."""
import ctypes
import gc

import numpy as np
import pytest

from pyc2ray_amd import _pinned


class MallocLib:
    def __init__(self):
        self.libc = ctypes.CDLL(None)
        self.libc.malloc.restype = ctypes.c_void_p
        self.libc.malloc.argtypes = [ctypes.c_size_t]
        self.libc.free.argtypes = [ctypes.c_void_p]
        self.allocs = self.frees = 0
        self.fail = False

    def asora_host_alloc(self, n, pp):
        if self.fail:
            return 2
        ctypes.cast(pp, ctypes.POINTER(ctypes.c_void_p))[0] = self.libc.malloc(n)
        self.allocs += 1
        return 0

    def asora_host_free(self, p):
        self.libc.free(p)
        self.frees += 1
        return 0


@pytest.fixture
def lib():
    lib = MallocLib()
    yield lib
    gc.collect()
    _pinned.release_free_buffers(lib)
    assert _pinned.stats() == {"pinned_bytes": 0, "free": {}}
    assert lib.allocs == lib.frees


@pytest.mark.parametrize("order", ["C", "F"])
def test_arrays_look_like_numpy_empty(lib, order):
    a = _pinned.empty(lib, (5, 5, 5), order=order)
    assert a.shape == (5, 5, 5) and a.dtype == np.float64 and a.flags.writeable and a.flags.aligned
    assert a.flags.f_contiguous == (order == "F") and a.flags.c_contiguous == (order == "C")
    a[...] = np.arange(125.0).reshape(5, 5, 5)
    assert a[1, 2, 3] == 38.0 and a.copy().flags.owndata


def test_a_buffer_is_reused_only_after_the_last_view_is_gone(lib):
    a = _pinned.empty(lib, (4, 4, 4))
    a[...] = 7.0
    view = a[1:3].T
    address = a.ctypes.data
    del a
    gc.collect()
    assert _pinned.stats()["free"] == {} and view.sum() == 7.0 * 32          # still owned by the view
    b = _pinned.empty(lib, (4, 4, 4))
    assert b.ctypes.data != address and lib.allocs == 2
    del view
    gc.collect()
    assert _pinned.stats()["free"] == {512: 1}
    c = _pinned.empty(lib, (4, 4, 4), order="F")
    assert c.ctypes.data == address and lib.allocs == 2


def test_caps_and_failures_fall_back_to_pageable_arrays(lib, monkeypatch):
    monkeypatch.setattr(_pinned, "MAX_PINNED_BYTES", 2 * 512)
    held = [_pinned.empty(lib, (4, 4, 4)) for _ in range(4)]
    assert lib.allocs == 2 and [h.flags.owndata for h in held] == [False, False, True, True]
    del held
    gc.collect()
    monkeypatch.setattr(_pinned, "MAX_PINNED_BYTES", 1 << 30)
    lib.fail = True
    x = _pinned.empty(lib, (3, 3, 3))                  # other size: needs a new buffer, which the library refuses
    assert x.flags.owndata and _pinned.stats()["pinned_bytes"] == 2 * 512
    lib.fail = False


def test_free_list_is_bounded(lib):
    many = [_pinned.empty(lib, (2, 2, 2)) for _ in range(_pinned.MAX_FREE_PER_SIZE + 3)]
    del many
    gc.collect()
    assert _pinned.stats()["free"] == {64: _pinned.MAX_FREE_PER_SIZE} and lib.frees == 3
