"""Initialisation guards of the raytracing library, as pyc2ray/asora_core.py:9-57.

Same four functions, same RuntimeError behaviour when used before device_init.  The flag is still
called ``cuda_init`` so that code written against the reference keeps working; the device is an
MI355X driven through HIP.
"""
from .load_extensions import load_asora

__all__ = ['cuda_is_init', 'device_init', 'device_close', 'photo_table_to_device']

# Whether device memory has been allocated (asora_core.py:14)
cuda_init = False


def cuda_is_init():
    return cuda_init


def device_init(N, source_batch_size, device_id=None):
    """Initialise the GPU and allocate the grid memory (asora_core.py:20-37).

    source_batch_size is accepted for compatibility: this build keeps the per-source column-density
    scratch in LDS, so no batch-sized N^3 slab exists.  device_id (extension) selects the GPU of this
    process; default is the current device, as in the reference (src/asora/memory.cu:39)."""
    global cuda_init
    libasora = load_asora()
    if libasora is not None:
        libasora.device_init(N, source_batch_size, device_id)
        cuda_init = True
    else:  # pragma: no cover - load_asora raises instead of returning None in this build
        raise RuntimeError("Could not initialize GPU: ASORA library not loaded")


def device_close():
    """Deallocate GPU memory (asora_core.py:39-47)."""
    global cuda_init
    if cuda_init:
        load_asora().device_close()
        cuda_init = False
    else:
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")


def photo_table_to_device(thin_table, thick_table):
    """Copy the optically thin & thick radiation tables to the GPU (asora_core.py:49-57).
    NumTau passed down = number of table elements, as in the reference (asora_core.py:54)."""
    NumTau = thin_table.shape[0]
    if cuda_init:
        load_asora().photo_table_to_device(thin_table, thick_table, NumTau)
    else:
        raise RuntimeError("GPU not initialized. Please initialize it by calling device_init(N)")
