"""Oracle vs the compiled reference (oracle/_ref) on randomised cases.  Build container only
(the reference sources are not on the GPU box); skipped when oracle/_ref is absent."""
import numpy as np
import pytest

import aad_amd
import oracle_binding as ob
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def ref():
    return aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))


@pytest.mark.parametrize("kind", ["music", "noise", "nyquist"])
@pytest.mark.parametrize("bits", [4, 3, 2])
def test_random_streams(ref, kind, bits):
    rng = np.random.default_rng(bits * 7 + len(kind))
    for i in range(12):
        ch = int(rng.integers(1, 3))
        ms = bool(ch == 2 and rng.integers(0, 2))
        trials = int(rng.integers(0, 3))
        mbs = int(rng.choice([64, 128, 200, 256, 1024, 4096]))
        n = int(rng.integers(1, 6000))
        pcm = synth_pcm(1, n, ch, seed=1000 + i, kind=kind)[0]
        a = ref.encode(pcm, bits, mbs, 44100, ms, trials)
        assert ob.encode(pcm, bits, mbs, 44100, ms, trials) == a, (ch, ms, trials, mbs, n)
        da, _ = ref.decode(a)
        db, _ = ob.decode(a)
        assert np.array_equal(da, db)


def test_state_carries_across_calls(ref):
    """A reused handle keeps its weights (src/aad_encoder.c:853-886; SetEncodeParameter only resets
    the step index, :797-799)."""
    import ctypes as C
    lib = ref.lib
    enc = lib.AADEncoder_Create(1024, None, 0)
    lanes = ob.fresh_lanes(2)
    try:
        for k in range(3):
            pcm = synth_pcm(1, 2500 + k * 17, 2, seed=50 + k)[0]
            a = ref.encode(pcm, 4, 1024, 48000, False, 1, encoder=enc)
            b = ob.encode(pcm, 4, 1024, 48000, False, 1, lanes=lanes, reset_idx=True)
            assert a == b, k
    finally:
        lib.AADEncoder_Destroy(enc)


def test_reference_cli_reproduces_fixtures(tmp_path):
    """BASELINE config 1: the reference CLI on its own inputs (plumbing, CPU path)."""
    import os
    import subprocess
    fix = os.path.join(os.path.dirname(__file__), "golden", "ref_fixtures")
    for name in ("sin300Hz_mono", "sin300Hz"):
        out = tmp_path / (name + ".aad")
        subprocess.run([ob.REF_CLI, "-e", os.path.join(fix, name + ".wav"), str(out)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert out.read_bytes() == open(os.path.join(fix, name + ".aad"), "rb").read()
        wav = tmp_path / (name + ".wav")
        subprocess.run([ob.REF_CLI, "-d", str(out), str(wav)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert wav.read_bytes() == open(os.path.join(fix, name + "_decoded.wav"), "rb").read()
