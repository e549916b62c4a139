"""ctypes view of the C-ABI in include/*.h.

``LegacyCodec`` drives the reference-shaped API (AADEncoder_* / AADDecoder_*, reference
src/aad_encoder.h:25-50, src/aad_decoder.h:15-42) of ANY library exporting it: this repo's
libaad_hip.so in production, the compiled reference (oracle/_ref/libaadref.so) in the tests -
the same call sequence the reference CLI uses (src/main.c:91-106, 182-198).
"""
import ctypes as C
import os

import numpy as np


class AADApiResult:
    OK = 0
    INVALID_ARGUMENT = 1
    INVALID_FORMAT = 2
    INSUFFICIENT_BUFFER = 3
    INSUFFICIENT_DATA = 4
    PARAMETER_NOT_SET = 5
    NG = 6
    _names = {0: "OK", 1: "INVALID_ARGUMENT", 2: "INVALID_FORMAT", 3: "INSUFFICIENT_BUFFER",
              4: "INSUFFICIENT_DATA", 5: "PARAMETER_NOT_SET", 6: "NG"}

    @classmethod
    def name(cls, v):
        return cls._names.get(int(v), str(v))


class AADHeaderInfo(C.Structure):  # include/aad.h
    _fields_ = [("format_version", C.c_uint32), ("codec_version", C.c_uint32),
                ("num_channels", C.c_uint16), ("num_samples", C.c_uint32),
                ("sampling_rate", C.c_uint32), ("bits_per_sample", C.c_uint16),
                ("block_size", C.c_uint16), ("num_samples_per_block", C.c_uint32),
                ("ch_process_method", C.c_int)]


class AADEncodeParameter(C.Structure):  # include/aad_encoder.h
    _fields_ = [("num_channels", C.c_uint16), ("sampling_rate", C.c_uint32),
                ("bits_per_sample", C.c_uint16), ("max_block_size", C.c_uint16),
                ("ch_process_method", C.c_int), ("num_encode_trials", C.c_uint8)]


class AADHipStreamDesc(C.Structure):  # include/aad_hip.h
    _fields_ = [("pcm_offset", C.c_uint64), ("data_offset", C.c_uint64),
                ("data_size", C.c_uint64), ("num_samples", C.c_uint32), ("reserved", C.c_uint32)]


class AADHipLaneState(C.Structure):  # include/aad_hip.h
    _fields_ = [("weight", C.c_int32 * 4), ("history", C.c_int32 * 4),
                ("stepsize_index", C.c_int32), ("quantize_error", C.c_int32)]


STREAM_DESC_DTYPE = np.dtype([("pcm_offset", "<u8"), ("data_offset", "<u8"), ("data_size", "<u8"),
                              ("num_samples", "<u4"), ("reserved", "<u4")])
ERROR_STATS_DTYPE = np.dtype([("rms_error", "<f8"), ("mean_abs_error", "<f8"), ("max_abs_error", "<f8")])  # AADHipErrorStats
RECONSTRUCT_DECODED, RECONSTRUCT_RESIDUAL = 0, 1  # enum AADHipReconstructOutput
OPTION_LANE_MAPPING, OPTION_TRIAL_LANES, OPTION_STAGING_THREADS, OPTION_TILE_KBYTES, OPTION_COMPARE_ORDER = 0, 1, 2, 3, 4  # enum AADHipOption
LANE_MAPPINGS = {"auto": 0, "dense": 1, "quad": 2, "quad-fused": 3, "dense-tiled": 4}  # enum AADHipLaneMapping
TRIAL_LANES = {"dual": 0, "single": 1}  # enum AADHipTrialLanes
LANE_STATE_DTYPE = np.dtype([("weight", "<i4", (4,)), ("history", "<i4", (4,)),
                             ("stepsize_index", "<i4"), ("quantize_error", "<i4")])

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.environ.get("AAD_HIP_LIBRARY", os.path.join(_PKG_DIR, "libaad_hip.so"))

LEGACY_SYMBOLS = [
    "AADEncoder_CalculateBlockSize", "AADEncoder_EncodeHeader", "AADEncoder_CalculateWorkSize",
    "AADEncoder_Create", "AADEncoder_Destroy", "AADEncoder_SetEncodeParameter", "AADEncoder_EncodeWhole",
    "AADDecoder_DecodeHeader", "AADDecoder_CalculateWorkSize", "AADDecoder_Create", "AADDecoder_Destroy",
    "AADDecoder_SetHeader", "AADDecoder_DecodeBlock", "AADDecoder_DecodeWhole",
]
HIP_SYMBOLS = [
    "AADHip_GetDeviceCount", "AADHip_ContextCreate", "AADHip_ContextDestroy", "AADHip_ContextSynchronize",
    "AADHip_ContextLastError", "AADHip_ContextSignalNextRun", "AADHip_SignalNextRunSupported", "AADHip_ContextSetOption", "AADHip_CalculateEncodedSize", "AADHip_EncodePlanCreate",
    "AADHip_EncodePlanDestroy", "AADHip_EncodePlanRun", "AADHip_DecodePlanCreate", "AADHip_DecodePlanDestroy",
    "AADHip_DecodePlanRun", "AADHip_EncodeBatch", "AADHip_DecodeBatch",
    "AADHip_ReconstructPlanCreate", "AADHip_ReconstructPlanDestroy", "AADHip_ReconstructPlanRun",
    "AADHip_ReconstructBatch",
]
WAV_SYMBOLS = ["AADWav_ParseHeader", "AADWav_WriteHeader", "AADWav_ConvertToPcm16"]
SYNTH_SYMBOLS = ["AADSynth_Generate"]


class AADWavInfo(C.Structure):  # include/aad_wav.h
    _fields_ = [("format_tag", C.c_uint16), ("num_channels", C.c_uint16), ("sampling_rate", C.c_uint32),
                ("bits_per_sample", C.c_uint16), ("num_samples", C.c_uint32), ("data_offset", C.c_uint64),
                ("data_size", C.c_uint64)]


def _declare_legacy(lib):
    u8p, i32pp = C.POINTER(C.c_uint8), C.POINTER(C.POINTER(C.c_int32))
    lib.AADEncoder_CalculateBlockSize.argtypes = [C.c_uint16, C.c_uint16, C.c_uint32,
                                                  C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    lib.AADEncoder_CalculateBlockSize.restype = C.c_int
    lib.AADEncoder_EncodeHeader.argtypes = [C.POINTER(AADHeaderInfo), u8p, C.c_uint32]
    lib.AADEncoder_EncodeHeader.restype = C.c_int
    lib.AADEncoder_CalculateWorkSize.argtypes = [C.c_uint16]
    lib.AADEncoder_CalculateWorkSize.restype = C.c_int32
    lib.AADEncoder_Create.argtypes = [C.c_uint16, C.c_void_p, C.c_int32]
    lib.AADEncoder_Create.restype = C.c_void_p
    lib.AADEncoder_Destroy.argtypes = [C.c_void_p]
    lib.AADEncoder_Destroy.restype = None
    lib.AADEncoder_SetEncodeParameter.argtypes = [C.c_void_p, C.POINTER(AADEncodeParameter)]
    lib.AADEncoder_SetEncodeParameter.restype = C.c_int
    lib.AADEncoder_EncodeWhole.argtypes = [C.c_void_p, i32pp, C.c_uint32, u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.AADEncoder_EncodeWhole.restype = C.c_int
    lib.AADDecoder_DecodeHeader.argtypes = [u8p, C.c_uint32, C.POINTER(AADHeaderInfo)]
    lib.AADDecoder_DecodeHeader.restype = C.c_int
    lib.AADDecoder_CalculateWorkSize.argtypes = []
    lib.AADDecoder_CalculateWorkSize.restype = C.c_int32
    lib.AADDecoder_Create.argtypes = [C.c_void_p, C.c_int32]
    lib.AADDecoder_Create.restype = C.c_void_p
    lib.AADDecoder_Destroy.argtypes = [C.c_void_p]
    lib.AADDecoder_Destroy.restype = None
    lib.AADDecoder_SetHeader.argtypes = [C.c_void_p, C.POINTER(AADHeaderInfo)]
    lib.AADDecoder_SetHeader.restype = C.c_int
    lib.AADDecoder_DecodeBlock.argtypes = [C.c_void_p, u8p, C.c_uint32, i32pp, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_uint32)]
    lib.AADDecoder_DecodeBlock.restype = C.c_int
    lib.AADDecoder_DecodeWhole.argtypes = [C.c_void_p, u8p, C.c_uint32, i32pp, C.c_uint32, C.c_uint32]
    lib.AADDecoder_DecodeWhole.restype = C.c_int


def _declare_hip(lib):
    vp = C.c_void_p
    lib.AADHip_GetDeviceCount.argtypes = []
    lib.AADHip_GetDeviceCount.restype = C.c_int32
    lib.AADHip_ContextCreate.argtypes = [C.c_int32, vp, C.POINTER(vp)]
    lib.AADHip_ContextCreate.restype = C.c_int
    lib.AADHip_ContextDestroy.argtypes = [vp]
    lib.AADHip_ContextDestroy.restype = None
    lib.AADHip_ContextSynchronize.argtypes = [vp]
    lib.AADHip_ContextSynchronize.restype = C.c_int
    lib.AADHip_ContextLastError.argtypes = [vp]
    lib.AADHip_ContextLastError.restype = C.c_char_p
    lib.AADHip_ContextSignalNextRun.argtypes = [vp, vp, vp]
    lib.AADHip_ContextSignalNextRun.restype = C.c_int
    lib.AADHip_SignalNextRunSupported.argtypes = []
    lib.AADHip_SignalNextRunSupported.restype = C.c_int32
    lib.AADHip_ContextSetOption.argtypes = [vp, C.c_int32, C.c_int32]
    lib.AADHip_ContextSetOption.restype = C.c_int
    lib.AADHip_CalculateEncodedSize.argtypes = [C.POINTER(AADEncodeParameter), C.c_uint32]
    lib.AADHip_CalculateEncodedSize.restype = C.c_uint64
    lib.AADHip_EncodePlanCreate.argtypes = [vp, C.POINTER(AADEncodeParameter), C.c_uint32, vp, C.POINTER(vp)]
    lib.AADHip_EncodePlanCreate.restype = C.c_int
    lib.AADHip_EncodePlanDestroy.argtypes = [vp]
    lib.AADHip_EncodePlanDestroy.restype = None
    lib.AADHip_EncodePlanRun.argtypes = [vp, vp, vp, vp]
    lib.AADHip_EncodePlanRun.restype = C.c_int
    lib.AADHip_DecodePlanCreate.argtypes = [vp, C.POINTER(AADHeaderInfo), C.c_int32, C.c_uint32, vp, C.POINTER(vp)]
    lib.AADHip_DecodePlanCreate.restype = C.c_int
    lib.AADHip_DecodePlanDestroy.argtypes = [vp]
    lib.AADHip_DecodePlanDestroy.restype = None
    lib.AADHip_DecodePlanRun.argtypes = [vp, vp, vp]
    lib.AADHip_DecodePlanRun.restype = C.c_int
    lib.AADHip_EncodeBatch.argtypes = [vp, C.POINTER(AADEncodeParameter), C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.AADHip_EncodeBatch.restype = C.c_int
    lib.AADHip_DecodeBatch.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp]
    lib.AADHip_DecodeBatch.restype = C.c_int
    lib.AADHip_ReconstructPlanCreate.argtypes = [vp, C.POINTER(AADEncodeParameter), C.c_uint32, vp, C.POINTER(vp)]
    lib.AADHip_ReconstructPlanCreate.restype = C.c_int
    lib.AADHip_ReconstructPlanDestroy.argtypes = [vp]
    lib.AADHip_ReconstructPlanDestroy.restype = None
    lib.AADHip_ReconstructPlanRun.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
    lib.AADHip_ReconstructPlanRun.restype = C.c_int
    lib.AADHip_ReconstructBatch.argtypes = [vp, C.POINTER(AADEncodeParameter), C.c_uint32, vp, vp, C.c_int32, vp, vp]
    lib.AADHip_ReconstructBatch.restype = C.c_int
    lib.AADWav_ParseHeader.argtypes = [vp, C.c_uint64, C.POINTER(AADWavInfo)]
    lib.AADWav_ParseHeader.restype = C.c_int
    lib.AADWav_WriteHeader.argtypes = [vp, C.c_uint32, C.c_uint16, C.c_uint32, C.c_uint32]
    lib.AADWav_WriteHeader.restype = C.c_int
    lib.AADWav_ConvertToPcm16.argtypes = [vp, C.c_uint16, C.c_uint64, vp]
    lib.AADWav_ConvertToPcm16.restype = C.c_int


def load_library(path=None, hip=True):
    """Load a library exporting the AAD C API.  The product library must exist: there is no
    fallback of any kind (build it with ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    path = path or LIBRARY_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s is missing - the HIP extension has not been built (run __graft_entry__.build())" % path)
    if hip:
        # torch ships a HIP runtime of its own.  Whichever libamdhip64 is mapped first serves both; loaded
        # the other way round (this library first, torch later) the process ends up with two runtimes and
        # AADHip_ContextCreate fails - so python callers that have torch get it imported here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    _declare_legacy(lib)
    if hip:
        _declare_hip(lib)
    return lib


def make_parameter(channels, bits, max_block_size=1024, rate=48000, ms=False, trials=0):
    return AADEncodeParameter(num_channels=channels, sampling_rate=rate, bits_per_sample=bits,
                              max_block_size=max_block_size, ch_process_method=1 if ms else 0,
                              num_encode_trials=trials)


def _planar_pointers(planar):
    """planar: int32 array [channels, samples] (C-contiguous) -> int32** for the legacy API"""
    rows = (C.POINTER(C.c_int32) * planar.shape[0])()
    for c in range(planar.shape[0]):
        rows[c] = planar[c].ctypes.data_as(C.POINTER(C.c_int32))
    return rows


class ApiError(RuntimeError):
    def __init__(self, where, code):
        super().__init__("%s -> %s" % (where, AADApiResult.name(code)))
        self.code = int(code)


class LegacyCodec:
    """The reference's call sequences (src/main.c:91-106, 182-198) against a loaded library."""

    def __init__(self, lib):
        self.lib = lib

    def block_size(self, max_block_size, channels, bits):
        bs, spb = C.c_uint16(0), C.c_uint32(0)
        rc = self.lib.AADEncoder_CalculateBlockSize(max_block_size, channels, bits, C.byref(bs), C.byref(spb))
        return rc, bs.value, spb.value

    def encode(self, pcm, bits=4, max_block_size=1024, rate=48000, ms=False, trials=0, encoder=None,
               capacity=None):
        """pcm: int16/int32 array [samples, channels] -> bytes of the .aad image."""
        pcm = np.asarray(pcm)
        n, ch = pcm.shape
        planar = np.ascontiguousarray(pcm.T.astype(np.int32))
        param = make_parameter(ch, bits, max_block_size, rate, ms, trials)
        own = encoder is None
        if own:
            encoder = self.lib.AADEncoder_Create(max_block_size, None, 0)
            if not encoder:
                raise ApiError("AADEncoder_Create", AADApiResult.NG)
        try:
            rc = self.lib.AADEncoder_SetEncodeParameter(encoder, C.byref(param))
            if rc != 0:
                raise ApiError("AADEncoder_SetEncodeParameter", rc)
            cap = int(capacity) if capacity is not None else max(64, n * ch * 2 + 64 + 18 * ch * (n // 4 + 2))
            out = np.zeros(cap, dtype=np.uint8)
            size = C.c_uint32(0)
            rc = self.lib.AADEncoder_EncodeWhole(encoder, _planar_pointers(planar), n,
                                                 out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(size))
            if rc != 0:
                raise ApiError("AADEncoder_EncodeWhole", rc)
            return out[:size.value].tobytes()
        finally:
            if own:
                self.lib.AADEncoder_Destroy(encoder)

    def decode_header(self, data):
        buf = np.frombuffer(data, dtype=np.uint8)
        h = AADHeaderInfo()
        rc = self.lib.AADDecoder_DecodeHeader(buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf), C.byref(h))
        if rc != 0:
            raise ApiError("AADDecoder_DecodeHeader", rc)
        return h

    def decode(self, data):
        """bytes of an .aad image -> (int16 array [samples, channels], header)"""
        h = self.decode_header(data)
        buf = np.frombuffer(data, dtype=np.uint8)
        planar = np.zeros((h.num_channels, h.num_samples), dtype=np.int32)
        dec = self.lib.AADDecoder_Create(None, 0)
        try:
            rc = self.lib.AADDecoder_DecodeWhole(dec, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf),
                                                 _planar_pointers(planar), h.num_channels, h.num_samples)
            if rc != 0:
                raise ApiError("AADDecoder_DecodeWhole", rc)
        finally:
            self.lib.AADDecoder_Destroy(dec)
        return np.ascontiguousarray(planar.T).astype(np.int16), h

    def decode_block(self, header, block, want_samples):
        buf = np.frombuffer(block, dtype=np.uint8)
        planar = np.zeros((header.num_channels, max(want_samples, 1)), dtype=np.int32)
        got = C.c_uint32(0)
        dec = self.lib.AADDecoder_Create(None, 0)
        try:
            rc = self.lib.AADDecoder_SetHeader(dec, C.byref(header))
            if rc != 0:
                raise ApiError("AADDecoder_SetHeader", rc)
            rc = self.lib.AADDecoder_DecodeBlock(dec, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf),
                                                 _planar_pointers(planar), header.num_channels, want_samples,
                                                 C.byref(got))
            if rc != 0:
                raise ApiError("AADDecoder_DecodeBlock", rc)
        finally:
            self.lib.AADDecoder_Destroy(dec)
        return np.ascontiguousarray(planar[:, :got.value].T).astype(np.int16)
