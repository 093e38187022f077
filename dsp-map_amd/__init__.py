"""dsp-map_amd: MI355X-native particle-based dynamic occupancy mapper.

Only what the hot path needs: csrc/ (hand-written gfx950 HIP kernels + the C
ABI of include/dspmap.h), capi.py (ctypes mirror of the reference's DSPMap class
surface), scene.py (synthetic depth-camera stream for benchmarks/tests) and
sharded.py (Z-slab multi-GPU driver over torch.distributed).

The directory name contains a hyphen (it is fixed by the project layout), so
import it through the `dsp_map_amd` shim at the repository root.
"""
from . import capi  # noqa: F401
from .capi import DSPMap, make_config, load_library, VPOINT_DTYPE  # noqa: F401
