"""Life cycle of the device state: the four functions of pyc2ray/asora_core.py:9-57 with the same names and the
same RuntimeError when something is used before ``device_init``.

The query is still called ``cuda_is_init`` so that scripts written for the reference run unchanged; the device
behind it is an MI355X driven through HIP.
"""
from . import _pinned, _residency
from .load_extensions import load_asora

__all__ = ['cuda_is_init', 'device_init', 'device_close', 'photo_table_to_device']

_NOT_READY = "GPU not initialized. Please initialize it by calling device_init(N)"


class _Lifecycle:
    """One flag: whether this process holds device grids (the reference's module-level ``cuda_init``)."""
    ready = False

    @classmethod
    def library(cls, need_ready=True):
        if need_ready and not cls.ready:
            raise RuntimeError(_NOT_READY)
        return load_asora()


def cuda_is_init():
    return _Lifecycle.ready


def device_init(N, source_batch_size, device_id=None):
    """Allocate the grids for an N^3 mesh on the GPU.

    ``source_batch_size`` is accepted and ignored: the live column densities of a source sit in LDS, so there is no
    batch-sized N^3 scratch to size.  ``device_id`` (extension) picks the GPU of this process; by default the
    current device is used, as the reference does (src/asora/memory.cu:39)."""
    if _Lifecycle.ready:
        _residency.reclaim()          # a re-initialisation drops the device grids: resident C2Ray objects fetch theirs first
    _Lifecycle.library(need_ready=False).device_init(N, source_batch_size, device_id)
    _Lifecycle.ready = True


def device_close():
    """Release the device grids (and the page-locked buffers no result array uses any more)."""
    lib = _Lifecycle.library()
    _residency.reclaim()              # results that exist only on the device (C2Ray.device_resident) come home first
    lib.device_close()
    _pinned.release_free_buffers(lib._lib)
    _Lifecycle.ready = False


def photo_table_to_device(thin_table, thick_table):
    """Upload the optically thin and thick photo-ionisation tables.  The library is told the number of table
    elements, which is what the reference passes as NumTau (asora_core.py:54)."""
    _Lifecycle.library().photo_table_to_device(thin_table, thick_table, thin_table.shape[0])
