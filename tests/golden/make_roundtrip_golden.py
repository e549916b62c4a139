#!/usr/bin/env python3
"""Golden results for the reference's own encode->decode suite (test/test_aad_encode_decode.c:283-616).

That suite pins nothing bit-exactly: it checks an RMSE bound per case.  Here the same inputs and
the same parameter grid are pushed through the COMPILED reference (oracle/_ref/libaadref.so, built
by oracle/Makefile from the reference's own sources) and what it produced is recorded:

  roundtrip_inputs.npz   the suite's three synthetic inputs as it builds them (440 Hz half-scale
                         sine; glibc srand(0)/rand() full-scale noise; Nyquist square) - int16
                         [2048, 2], kept as data so no libc/libm detail leaks into the tests
  ref_fixtures/bunny1.wav, pi_15-25sec.wav
                         the two real-audio data files of the reference's test directory
  roundtrip_suite.json   per case: parameters, the suite's RMSE bound, the RMSE the reference
                         reaches, SHA-256 of its .aad image and of its decoded PCM

Runs only in the build container.  Nothing of the reference's source is copied.
"""
import ctypes
import itertools
import json
import math
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import aad_amd  # noqa: E402
from helpers import read_wav16, sha256  # noqa: E402

REF_TEST_DIR = "/root/reference/test"
INT16_MAX = 32767
EPS_SYNTH = {"sine": {4: 5.0e-2, 3: 6.0e-2, 2: 8.0e-2}, "white_noise": {4: 1.0e-1, 3: 1.5e-1, 2: 2.4e-1},
             "nyquist": {4: 1.2e-1, 3: 1.6e-1, 2: 2.3e-1}}
EPS_WAV = {4: 5.0e-2, 3: 6.0e-2, 2: 8.0e-2}
WAV_FILES = ["unit_impulse_mono.wav", "unit_impulse.wav", "sin300Hz_mono.wav", "sin300Hz.wav", "bunny1.wav",
             "pi_15-25sec.wav"]


def synthetic_inputs():
    n, ch = 2048, 2
    sine = np.array([[int(INT16_MAX * 0.5 * math.sin((2.0 * 3.1415 * 440.0 * s) / 48000.0)) for _ in range(ch)]
                     for s in range(n)], dtype=np.int16)
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(0)
    rand_max = 2147483647
    noise = np.zeros((n, ch), dtype=np.int16)
    for c in range(ch):          # channel-major draw order, as the suite fills its rows
        for s in range(n):
            f32 = np.float32
            noise[s, c] = int(float(f32(INT16_MAX) * f32(2.0)) * (libc.rand() / rand_max - 0.5))
    nyq = np.array([[-32768 if s % 2 else 32767] * ch for s in range(n)], dtype=np.int16)
    return {"sine": sine, "white_noise": noise, "nyquist": nyq}


def suite_rmse(x, y):
    """the bound the suite checks (test/test_aad_encode_decode.c:246-259)"""
    d = x.astype(np.float64) / INT16_MAX - y.astype(np.float64) / INT16_MAX
    return float(np.sqrt(np.sum(d * d) / d.size))


def main():
    ref = aad_amd.LegacyCodec(aad_amd.load_library(os.path.join(ROOT, "oracle", "_ref", "libaadref.so"), hip=False))
    for f in ("bunny1.wav", "pi_15-25sec.wav"):
        shutil.copyfile(os.path.join(REF_TEST_DIR, f), os.path.join(HERE, "ref_fixtures", f))
    synth = synthetic_inputs()
    np.savez_compressed(os.path.join(HERE, "roundtrip_inputs.npz"), **synth)

    cases = []

    def run(name, pcm, rate, bits, mbs, ms, trials, eps, file_bytes=None):
        aad = ref.encode(pcm, bits, mbs, rate, ms, trials)
        dec, _ = ref.decode(aad)
        rmse = suite_rmse(pcm, dec)
        assert rmse < eps, (name, bits, mbs, ms, trials, rmse)
        if file_bytes is not None:
            assert len(aad) < file_bytes // 2   # "must at least halve" (test/test_aad_encode_decode.c:236-239)
        cases.append(dict(input=name, channels=int(pcm.shape[1]), sampling_rate=rate, bits=bits, max_block_size=mbs,
                          ms=ms, trials=trials, rms_epsilon=eps, reference_rmse=rmse, aad_bytes=len(aad),
                          aad_sha256=sha256(aad), decoded_sha256=sha256(dec.tobytes())))

    for kind, pcm in synth.items():
        for trials, bits, mbs, (ch, ms) in itertools.product((0, 1), (4, 3, 2), (128, 1024),
                                                             ((1, False), (2, False), (2, True))):
            run(kind, pcm[:, :ch], 8000, bits, mbs, ms, trials, EPS_SYNTH[kind][bits])
    for f in WAV_FILES:
        path = os.path.join(HERE, "ref_fixtures", f)
        pcm, rate = read_wav16(path)
        size = os.path.getsize(path)
        for bits, mbs in itertools.product((4, 3, 2), (128, 256, 1024, 4096)):
            for ms in ((False, True) if pcm.shape[1] == 2 else (False,)):
                run(f, pcm, rate, bits, mbs, ms, 0, EPS_WAV[bits], size)
        run(f, pcm, rate, 4, 1024, False, 2, EPS_WAV[4], size)   # the CLI defaults on real audio
    with open(os.path.join(HERE, "roundtrip_suite.json"), "w") as fo:
        json.dump(dict(generator="tests/golden/make_roundtrip_golden.py", cases=cases), fo, indent=1)
    print("round-trip suite cases:", len(cases))


if __name__ == "__main__":
    main()
