"""pyc2ray_amd -- MI355X-native implementation of pyc2ray's raytracing + chemistry hot path.

The public names are those of the reference package for this path (pyc2ray/__init__.py:1-9):
evolve3D, do_raytracing, cuda_is_init / device_init / device_close / photo_table_to_device,
hydrogenODE, printlog, format_sources, ...  All arithmetic runs in hand-written HIP kernels
(pyc2ray_amd/csrc) behind the C-ABI of include/asora_hip.h; there is no CPU compute path.
"""
from .evolve import *          # noqa: F401,F403  evolve3D, evolve3D_MPI
from .asora_core import *      # noqa: F401,F403
from .raytracing import *      # noqa: F401,F403
from .chemistry import *       # noqa: F401,F403
from .utils import *           # noqa: F401,F403
from .radiation import *       # noqa: F401,F403
from .c2ray_base import *      # noqa: F401,F403
from .c2ray_test import *      # noqa: F401,F403
from . import evolve, raytracing, chemistry, asora_core, utils, dist   # noqa: F401
