"""Inputs and cases of the reference's encode->decode suite (test/test_aad_encode_decode.c:283-616)
as recorded by tests/golden/make_roundtrip_golden.py from the compiled reference."""
import json
import os

import numpy as np

from helpers import GOLDEN, read_wav16

INT16_MAX = 32767
CASES = json.load(open(os.path.join(GOLDEN, "roundtrip_suite.json")))["cases"]
_inputs = {}


def suite_input(case):
    """-> (pcm int16 [samples, channels], bytes of the source WAV file or None)"""
    name = case["input"]
    if name not in _inputs:
        if name.endswith(".wav"):
            path = os.path.join(GOLDEN, "ref_fixtures", name)
            _inputs[name] = (read_wav16(path)[0], os.path.getsize(path))
        else:
            with np.load(os.path.join(GOLDEN, "roundtrip_inputs.npz")) as z:
                for k in z.files:
                    _inputs[k] = (z[k], None)
    pcm, size = _inputs[name]
    return np.ascontiguousarray(pcm[:, :case["channels"]]), size


def suite_rmse(x, y):
    """the figure the suite bounds (test/test_aad_encode_decode.c:246-259)"""
    d = x.astype(np.float64) / INT16_MAX - y.astype(np.float64) / INT16_MAX
    return float(np.sqrt(np.sum(d * d) / d.size))
