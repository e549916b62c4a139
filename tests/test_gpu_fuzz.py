"""Randomised GPU-vs-oracle parity over the parameter space the fixed cases leave gaps in: odd
max_block_size values (every pack-unit remainder), 1-8 channels, ragged stream lengths from a
single sample up to a few blocks, trials 0-2, M/S, every lane mapping.  Seeded, so a failure
names a reproducible case.  Bar: bit-exact images and decodes."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import ApiError, make_parameter
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("mapping", ["dense", "quad", "quad-fused", "auto"])
@pytest.mark.parametrize("seed", range(8))
def test_random_parameter_sets(engine, mapping, seed):
    rng = np.random.default_rng(1000 + seed)
    engine.set_mapping(mapping)
    try:
        for _ in range(4):
            ch = int(rng.choice([1, 2, 2, 2, 3, 5, 8]))
            bits = int(rng.choice([4, 3, 2]))
            ms = bool(ch == 2 and rng.integers(0, 3) == 0)
            trials = int(rng.choice([0, 0, 1, 2]))
            mbs = int(rng.integers(18 * ch, 18 * ch + 40)) if rng.integers(0, 4) == 0 else int(rng.integers(18 * ch, 3000))
            rc, block_size, spb = ob.geometry(mbs, ch, bits)
            streams = int(rng.integers(1, 24))
            lengths = [int(rng.integers(1, 4 * max(spb, 1) + 50)) if rc == 0 else 100 for _ in range(streams)]
            kind = str(rng.choice(["music", "noise", "nyquist"]))
            pcms = [synth_pcm(1, n, ch, seed=int(rng.integers(0, 1 << 30)), kind=kind)[0] for n in lengths]
            param = make_parameter(ch, bits, mbs, 48000, ms, trials)
            try:
                want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
            except RuntimeError:
                # the oracle refuses (no block fits, or a block would carry no data - reference
                # src/aad_encoder.c:106-108, 170-172): the engine must refuse as well
                with pytest.raises(ApiError):
                    engine.encode_host(pcms, param)
                continue
            images = engine.encode_host(pcms, param)
            for i, (got, w) in enumerate(zip(images, want)):
                assert got == w, (seed, mapping, ch, bits, ms, trials, mbs, lengths[i])
            decoded = engine.decode_host(images)
            for i, (got, w) in enumerate(zip(decoded, want)):
                assert np.array_equal(got, ob.decode(w)[0]), (seed, mapping, ch, bits, ms, trials, mbs, lengths[i])
    finally:
        engine.set_mapping("auto")
