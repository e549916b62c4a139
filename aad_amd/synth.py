"""Integer-only synthetic PCM corpus (SURVEY.md section 8d "Synthetic PCM generator").

Bit-reproducible on any machine (no libm, no floats): every (stream, channel)
pair owns a xorshift64 generator; a sample is two triangle partials plus
triangular-ish noise, clipped to int16.  Used by bench.py, the golden-vector
script and the parity tests, so the GPU box can rebuild exactly the inputs the
reference was run on in the build container.

Layout returned: int16 array [streams, samples, channels] (C order) - i.e. each
stream is channel-interleaved PCM, the layout the device kernels read.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M32 = np.uint64(0xFFFFFFFF)
_M16 = np.uint64(0xFFFF)


def _xorshift(x):
    x ^= x << np.uint64(13)
    x ^= x >> np.uint64(7)
    x ^= x << np.uint64(17)
    return x


def _tri(phase):
    """16-bit triangle wave from a 32-bit phase accumulator, range [-32767, 32767]."""
    t = (phase >> np.uint64(16)).astype(np.int64)
    v = np.where(t < 32768, t, 65535 - t)
    return 2 * v - 32767


_KINDS = {"music": 0, "noise": 1, "nyquist": 2}  # enum AADSynthKind (include/aad_synth.h)
_native = None


def _native_generator():
    """AADSynth_Generate from libaad_hip.so (aad_amd/csrc/aad_synth.c, the same arithmetic in C) or
    False when the library has not been built."""
    global _native
    if _native is None:
        try:
            import ctypes as C
            from .capi import LIBRARY_PATH
            fn = C.CDLL(LIBRARY_PATH).AADSynth_Generate
            fn.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int32, C.c_uint64]
            fn.restype = C.c_int32
            _native = fn
        except (OSError, AttributeError):
            _native = False
    return _native


def synth_pcm(num_streams, num_samples, channels, seed=1234, rate=48000, kind="music", first_stream=0,
              native=True):
    """kind: "music" (two partials + noise, |x| <~ 0.63 FS), "noise" (full-scale white
    noise) or "nyquist" (full-scale square at fs/2) - the two stress shapes of
    reference test/test_aad_encode_decode.c:447-451, 467-470.  native=False forces the numpy
    form below (the specification); by default the C form of the same arithmetic is used."""
    if kind not in _KINDS:
        raise ValueError(kind)
    fn = _native_generator() if native else False
    if fn:
        out = np.empty((num_streams, num_samples, channels), dtype=np.int16)
        if out.size and fn(out.ctypes.data, num_streams, num_samples, channels, seed, rate, _KINDS[kind], first_stream) != 0:
            raise ValueError("AADSynth_Generate refused its arguments")
        return out
    s = np.arange(first_stream, first_stream + num_streams, dtype=np.uint64)[:, None]
    c = np.arange(channels, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        x = (_GOLDEN * np.uint64(seed + 1)) ^ ((s << np.uint64(8)) | c)
        x = np.where(x == 0, np.uint64(1), x)
        for _ in range(4):
            x = _xorshift(x)
        x = _xorshift(x)
        f1 = np.uint64(100) + x % np.uint64(901)
        x = _xorshift(x)
        f2 = np.uint64(1000) + x % np.uint64(5001)
        x = _xorshift(x)
        ph1 = x & _M32
        x = _xorshift(x)
        ph2 = x & _M32
        inc1 = (f1 << np.uint64(32)) // np.uint64(rate)
        inc2 = (f2 << np.uint64(32)) // np.uint64(rate)
        out = np.empty((num_streams, num_samples, channels), dtype=np.int16)
        for n in range(num_samples):
            x = _xorshift(x)
            if kind == "music":
                u = ((x & _M16) + ((x >> np.uint64(16)) & _M16) + ((x >> np.uint64(32)) & _M16)).astype(np.int64)
                noise = ((u - 98304) * 437) >> 14
                v = ((_tri(ph1) * 11469) >> 15) + ((_tri(ph2) * 6554) >> 15) + noise
                ph1 = (ph1 + inc1) & _M32
                ph2 = (ph2 + inc2) & _M32
            elif kind == "noise":
                v = (x & _M16).astype(np.int64) - 32768
            elif kind == "nyquist":
                v = np.full(x.shape, 32767 if (n & 1) == 0 else -32768, dtype=np.int64)
            else:
                raise ValueError(kind)
            out[:, n, :] = np.clip(v, -32768, 32767).astype(np.int16)
    return out
