"""aad_amd - MI355X-native AAD (Ayashi Adaptive Differential PCM) encode/decode engine.

The product is the C-ABI shared library ``aad_amd/libaad_hip.so`` (headers in ``include/``):
the reference's AADEncoder_* / AADDecoder_* API plus the batched AADHip_* API, implemented with
hand-written HIP kernels for gfx950.  This Python package is only the binding layer used by
the tests, bench.py and the multi-GPU batch driver; there is no Python or CPU codec in it.
"""
from .capi import (  # noqa: F401
    AADApiResult,
    AADEncodeParameter,
    AADHeaderInfo,
    AADHipLaneState,
    AADHipStreamDesc,
    ApiError,
    LegacyCodec,
    load_library,
    make_parameter,
)

__all__ = [
    "AADApiResult",
    "AADEncodeParameter",
    "AADHeaderInfo",
    "AADHipLaneState",
    "AADHipStreamDesc",
    "ApiError",
    "LegacyCodec",
    "load_library",
    "make_parameter",
]
