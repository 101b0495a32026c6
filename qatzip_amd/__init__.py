"""qatzip_amd — MI355X-native backend for the QATzip chunked compress/decompress hot path.

Layout: csrc/ (hand-written gfx950 HIP kernels + the C ABI), _lib.py (ctypes view of the
device-resident ABI, include/qzamd_device.h), build.py (in-tree hipcc build).
"""
from . import build  # noqa: F401
from ._lib import Context, DevBuf, QzdError, load, max_deflate_len  # noqa: F401
