"""ctypes binding of the parity oracle (oracle/aad_oracle.c).  TESTS ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libaad_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libaadref.so")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "aad")


class AadoLane(C.Structure):
    _fields_ = [("w", C.c_int32 * 4), ("h", C.c_int32 * 4), ("idx", C.c_int32), ("qerr", C.c_int32)]


class AadoHeader(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("format_version", "codec_version", "num_channels", "num_samples",
                                          "sampling_rate", "bits_per_sample", "block_size",
                                          "samples_per_block", "ch_process_method")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(ORACLE_SO)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        l.aado_step_table.restype = C.POINTER(C.c_uint16)
        l.aado_index_deltas.argtypes = [u32]
        l.aado_index_deltas.restype = C.POINTER(C.c_int16)
        l.aado_block_geometry.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
        l.aado_put_header.argtypes = [C.POINTER(AadoHeader), vp, sz]
        l.aado_get_header.argtypes = [vp, sz, C.POINTER(AadoHeader)]
        l.aado_check_header.argtypes = [C.POINTER(AadoHeader), u32]
        l.aado_encoded_size.argtypes = [u32, u32, u32, u32]
        l.aado_encoded_size.restype = sz
        l.aado_encode_stream.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32, C.POINTER(AadoLane), vp, sz,
                                         C.POINTER(sz)]
        l.aado_decode_stream.argtypes = [vp, sz, u32, vp, u32, C.POINTER(AadoHeader)]
        l.aado_decode_block.argtypes = [C.POINTER(AadoHeader), vp, sz, vp, u32, C.POINTER(u32)]
        l.aado_encode_step.argtypes = [C.POINTER(AadoLane), C.c_int32, u32]
        l.aado_encode_step.restype = u32
        l.aado_decode_step.argtypes = [C.POINTER(AadoLane), u32, u32]
        l.aado_decode_step.restype = C.c_int32
        l.aado_encode_batch.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32, u32, vp, sz]
        l.aado_decode_batch.argtypes = [vp, u32, sz, sz, vp, u32]
        l.aado_residual.argtypes = [vp, vp, sz, vp]
        l.aado_residual.restype = None
        l.aado_error_stats.argtypes = [vp, vp, u32, u32, C.POINTER(C.c_double)]
        l.aado_error_stats.restype = None
        _lib = l
    return _lib


def geometry(max_block_size, channels, bits):
    bs, spb = C.c_uint32(0), C.c_uint32(0)
    rc = lib().aado_block_geometry(max_block_size, channels, bits, C.byref(bs), C.byref(spb))
    return rc, bs.value, spb.value


def encoded_size(num_samples, channels, bits, max_block_size=1024):
    return lib().aado_encoded_size(num_samples, channels, bits, max_block_size)


def fresh_lanes(channels=8):
    return (AadoLane * channels)()


def encode(pcm, bits=4, max_block_size=1024, rate=48000, ms=False, trials=0, lanes=None, reset_idx=True):
    """pcm: int16 [samples, channels] -> bytes.  lanes: carried state (AadoLane array) or None."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    n, ch = pcm.shape
    if lanes is None:
        lanes = fresh_lanes(max(ch, 1))
    if reset_idx:
        for c in range(len(lanes)):
            lanes[c].idx = 0
    cap = encoded_size(n, ch, bits, max_block_size) + 64
    out = np.zeros(cap, dtype=np.uint8)
    got = C.c_size_t(0)
    rc = lib().aado_encode_stream(pcm.ctypes.data, n, ch, rate, bits, max_block_size, 1 if ms else 0, trials,
                                  lanes, out.ctypes.data, cap, C.byref(got))
    if rc != 0:
        raise RuntimeError("oracle encode rc=%d" % rc)
    return out[:got.value].tobytes()


def decode(data, max_channels=8):
    buf = np.frombuffer(data, dtype=np.uint8)
    hd = AadoHeader()
    rc = lib().aado_get_header(buf.ctypes.data, len(buf), C.byref(hd))
    if rc != 0:
        raise RuntimeError("oracle header rc=%d" % rc)
    pcm = np.zeros((hd.num_samples, hd.num_channels), dtype=np.int16)
    rc = lib().aado_decode_stream(buf.ctypes.data, len(buf), max_channels, pcm.ctypes.data, hd.num_samples, None)
    if rc != 0:
        raise RuntimeError("oracle decode rc=%d" % rc)
    return pcm, hd


def residual(x, y):
    """what `aad -g` writes for original x and reconstruction y (int16 [samples, channels])"""
    x = np.ascontiguousarray(x, dtype=np.int16)
    y = np.ascontiguousarray(y, dtype=np.int16)
    out = np.empty_like(x)
    lib().aado_residual(x.ctypes.data, y.ctypes.data, x.size, out.ctypes.data)
    return out


def error_stats(x, y):
    """(RMSE, MSD, MaxAE) exactly as `aad -c` computes them"""
    x = np.ascontiguousarray(x, dtype=np.int16)
    y = np.ascontiguousarray(y, dtype=np.int16)
    out = (C.c_double * 3)()
    lib().aado_error_stats(x.ctypes.data, y.ctypes.data, x.shape[0], x.shape[1], out)
    return tuple(out)


def stats_line(stats):
    """the line `aad -c` prints (src/main.c:493-497)"""
    return "RMSE:%f MSD:%f MaxAE:%f \n" % tuple(stats)
