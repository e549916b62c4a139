"""Encode-trial counts beyond 3 (num_encode_trials is a uint8_t; the reference's search loop takes any count,
src/aad_encoder.c:531-557).  The oracle against tests/golden/trials_high.json (images of the compiled reference for
t in {3, 4, 5, 7, 16, 255}; generator make_trials_golden.py), and - build container only - against the compiled
reference itself on fresh random cases with t up to 255."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, sha256

CASES = json.load(open(os.path.join(GOLDEN, "trials_high.json")))["cases"]


def case_pcm(c):
    return synth_pcm(1, c["num_samples"], c["channels"], seed=c["seed"], kind=c["kind"])[0]


def test_golden_spans_the_counts():
    assert {c["trials"] for c in CASES} == {3, 4, 5, 7, 16, 255} and len(CASES) == 108


@pytest.mark.parametrize("trials", [3, 4, 5, 7, 16, 255])
def test_oracle_matches_reference_images(trials):
    for c in (c for c in CASES if c["trials"] == trials):
        image = ob.encode(case_pcm(c), c["bits"], c["max_block_size"], 48000, c["ms"], trials)
        assert len(image) == c["aad_bytes"] and sha256(image) == c["aad_sha256"], c
        assert sha256(ob.decode(image)[0].astype("<i2").tobytes()) == c["decoded_sha256"], c


def test_more_trials_do_change_the_image():
    """the counts are not silently capped: somewhere in the set t = 4 / 7 / 16 pick a different processor than t = 3"""
    differ = 0
    for c in CASES[:40]:
        pcm = case_pcm(c)
        a = ob.encode(pcm, c["bits"], c["max_block_size"], 48000, c["ms"], 3)
        differ += any(ob.encode(pcm, c["bits"], c["max_block_size"], 48000, c["ms"], t) != a for t in (4, 7, 16))
    assert differ > 0


@pytest.mark.ref
def test_oracle_matches_compiled_reference_random_high_counts():
    import aad_amd
    ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
    rng = np.random.default_rng(31337)
    for i in range(60):
        ch = int(rng.integers(1, 3))
        bits = int(rng.choice([4, 3, 2]))
        ms = bool(ch == 2 and rng.integers(0, 2))
        trials = int(rng.choice([3, 4, 6, 9, 33, 100, 255]))
        mbs = int(rng.choice([64, 128, 200, 256, 1024]))
        n = int(rng.integers(1, 2500 if trials < 50 else 700))
        pcm = synth_pcm(1, n, ch, seed=2000 + i, kind=str(rng.choice(["music", "noise", "nyquist"])))[0]
        assert ob.encode(pcm, bits, mbs, 44100, ms, trials) == ref.encode(pcm, bits, mbs, 44100, ms, trials), (ch, bits, ms, trials, mbs, n)
