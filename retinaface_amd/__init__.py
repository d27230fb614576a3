"""retinaface_amd -- MI355X-native drop-in for the RetinaFace::detect() hot path of clancylian/retinaface.

The product is the C-ABI shared library ``retinaface_amd/lib/libretinaface_amd.so`` (hand-written gfx950 HIP
kernels + C++ host runtime, sources in ``retinaface_amd/csrc``, boundary in ``include/retinaface_amd.h``).
This package is a thin ctypes binding over that boundary plus the host-side helpers the benchmark and the
tests need (synthetic face-bearing frames, batch sharding for one-process-per-GPU runs).
There is no CPU fallback: importing works anywhere, but creating a detector without the built library or
without a HIP device raises.
"""
from ._lib import RFError, abi_version, lib_path, load_library  # noqa: F401
from .detector import PRECISION_FP16, PRECISION_FP32, PRECISION_INT8, Detection, RetinaFace  # noqa: F401
