"""ctypes binding of pyc2ray_amd/lib/libasora_hip.so (the C-ABI declared in include/asora_hip.h).

This is the only place the shared library is opened.  There is no fallback of any kind: when the
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYC2RAY_AMD_LIBASORA selects another BUILD of the same HIP library (diagnostic variants made by
# `make -C pyc2ray_amd/csrc EXTRA=... OUT=...`, tools/ab_prebuilt.sh): never a fallback -- a path that does not exist raises.
LIB_PATH = os.environ.get("PYC2RAY_AMD_LIBASORA") or os.path.join(_HERE, "lib", "libasora_hip.so")

# grid selectors / options / kernels, as in include/asora_hip.h
GRID_NDENS, GRID_XH_AV, GRID_PHI_ION, GRID_TEMP, GRID_XH, GRID_XH_INTERMED, GRID_PHI_HEAT = range(7)
(OPT_FORTRAN_CONSTANTS, OPT_GREY_NOTABLES, OPT_TIMING, OPT_Z_TRANSPOSED, OPT_BLOCK_THREADS, OPT_SECTORS,
 OPT_HEATING, OPT_C2RAY_OWN_FLUX, OPT_NO_UNIFORM_T, OPT_SUBBOX_GLOBAL_SHELLS, OPT_PIPELINED_COPIES,
 OPT_SKIP_ZERO_RATES, OPT_GLOBAL_ATOMICS, OPT_PAIR_SOURCES, OPT_SUBBOX_TABLES, OPT_ALIGNED_ROWS, OPT_GEOMETRY_ON_HOST,
 OPT_PLACEMENT_CANDIDATES) = range(18)
KERNEL_RAYTRACE, KERNEL_CHEMISTRY, KERNEL_PREP, KERNEL_FINISH = range(4)
VARIANT_PAIRED, VARIANT_ALIGNED, VARIANT_BUFFER_ATOMICS, VARIANT_SPLIT_DESCRIPTORS, VARIANT_SKIP_ZERO, VARIANT_GLOBAL_SHELLS = 1, 2, 4, 8, 16, 32

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

#: every symbol include/asora_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "asora_device_init": (C.c_int, [C.c_int, C.c_int]),
    "asora_device_init_ex": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "asora_device_close": (C.c_int, []),
    "asora_density_to_device": (C.c_int, [_dp, C.c_int]),
    "asora_photo_table_to_device": (C.c_int, [_dp, _dp, C.c_int]),
    "asora_heat_table_to_device": (C.c_int, [_dp, _dp, C.c_int]),
    "asora_source_data_to_device": (C.c_int, [_ip, _dp, C.c_int]),
    "asora_do_all_sources": (C.c_int, [C.c_double, _dp, C.c_double, C.c_double, _dp, _dp, _dp, C.c_int, C.c_int,
                                       C.c_double, C.c_double, C.c_int]),
    "c2ray_global_pass": (C.c_int, [C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "c2ray_do_all_sources": (C.c_int, [_dp, _ip, C.c_int, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp,
                                       C.c_float, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    "asora_last_error": (C.c_char_p, []),
    "asora_grid_to_device": (C.c_int, [C.c_int, _dp, C.c_int, C.c_char]),
    "asora_grid_to_host": (C.c_int, [C.c_int, _dp, C.c_int, C.c_char]),
    "asora_grid_copy": (C.c_int, [C.c_int, C.c_int]),
    "asora_grid_scale": (C.c_int, [C.c_int, C.c_double]),
    "asora_device_ptr": (C.c_void_p, [C.c_int]),
    "asora_grid_sum": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "asora_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "asora_host_free": (C.c_int, [C.c_void_p]),
    "asora_raytrace_device": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_int]),
    "asora_raytrace_begin": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "asora_raytrace_range": (C.c_int, [C.c_int, C.c_int]),
    "asora_raytrace_begin_planes": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _ip,
                                              C.c_int]),
    "asora_raytrace_fold": (C.c_int, [C.c_int, C.c_int]),
    "asora_stream": (C.c_void_p, []),
    "asora_subbox_raytrace_device": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double,
                                               C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    "asora_device_init_auto": (C.c_int, [C.c_int]),
    "asora_chemistry_device": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.POINTER(C.c_int), _dp, _dp]),
    "asora_chemistry_range": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.c_int, C.c_int, C.c_int]),
    "asora_chemistry_finish": (C.c_int, [C.POINTER(C.c_int), _dp, _dp]),
    "asora_reduction_ptr": (C.c_void_p, []),
    "asora_evolve_begin": (C.c_int, [C.c_double] * 11 + [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]),
    "asora_evolve_enqueue": (C.c_int, [C.c_int]),
    "asora_evolve_poll": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_int, C.POINTER(C.c_int)]),
    "asora_evolve_begin_slab": (C.c_int, [C.c_double] * 11 + [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]),
    "asora_evolve_slab_trace": (C.c_int, [C.c_int, C.c_int]),
    "asora_evolve_slab_fold_out": (C.c_int, [C.c_int, C.c_int]),
    "asora_evolve_slab_outbox": (C.c_void_p, []),
    "asora_evolve_slab_outbox_to_host": (C.c_int, [C.c_int, C.c_int, _dp]),
    "asora_evolve_slab_outbox_from_host": (C.c_int, [C.c_int, C.c_int, _dp]),
    "asora_evolve_slab_fold_all": (C.c_int, []),
    "asora_debug_placement": (None, [C.POINTER(C.c_int), _dp, _dp]),
    "asora_debug_init_cost": (None, [_dp, _dp]),
    "asora_evolve_slab_add": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "asora_evolve_slab_add_host": (C.c_int, [C.c_int, C.c_int, _dp]),
    "asora_evolve_slab_pass": (C.c_int, []),
    "asora_evolve_slab_nhi": (C.c_int, [C.c_int, C.c_int]),
    "asora_evolve_slab_close": (C.c_int, [_dp]),
    "asora_planes_to_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp]),
    "asora_planes_to_device": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp]),
    "asora_set_option": (C.c_int, [C.c_int, C.c_int]),
    "asora_get_option": (C.c_int, [C.c_int]),
    "asora_kernel_time_ms": (C.c_int, [C.c_int, _dp, C.POINTER(C.c_long)]),
    "asora_kernel_time_reset": (C.c_int, []),
    "asora_synchronize": (C.c_int, []),
    "asora_last_raytrace_counts": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "asora_last_raytrace_counts_ex": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "asora_debug_coldens": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, _dp, C.c_int]),
    "asora_last_raytrace_variant": (C.c_int, []),
    "asora_debug_geometry_bytes": (C.c_size_t, []),
    "asora_debug_geometry_table": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asora_build_id": (C.c_char_p, []),
    "asora_build_flags": (C.c_char_p, []),
}

_lib = None


def load():
    """Open libasora_hip.so once.  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"pyc2ray_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`make -C pyc2ray_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    # One HIP runtime per process.  torch bundles its own libamdhip64 (same SONAME as /opt/rocm's);
    # if torch were imported AFTER this library had pulled in /opt/rocm's copy, the process would
    # hold two runtimes and a device pointer of one would be unknown to the other (RCCL through
    # torch.distributed on our phi_ion grid).  Importing torch first makes both resolve to one copy.
    # PYC2RAY_AMD_NO_TORCH=1 skips this for torch-free single-GPU use.
    if os.environ.get("PYC2RAY_AMD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the build disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().asora_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else 'unknown error'}")


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)
