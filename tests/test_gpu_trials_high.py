"""Encode-trial counts beyond 3 on the GPU (reference src/aad_encoder.c:531-557 takes any uint8_t count): the golden images of
the compiled reference for t in {3, 4, 5, 7, 16, 255} (tests/golden/trials_high.json) through the host-memory batch entry point
under every encode lane mapping and BOTH trial-lane layouts ("dual": a second group of lanes encodes every candidate beside
the measuring chain, keeping at most two alternative encodes per stream - the slot logic that t > 2 exercises; "single": search
then encode on the same lanes), through the legacy API, and ragged same-parameter batches against the oracle, cut into tiles too."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import make_parameter
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, sha256

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "trials_high.json")))["cases"]


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def case_pcm(c):
    return synth_pcm(1, c["num_samples"], c["channels"], seed=c["seed"], kind=c["kind"])[0]


@pytest.mark.parametrize("trial_lanes", ["dual", "single"])
@pytest.mark.parametrize("mapping", ["auto", "dense", "quad"])
def test_golden_images_host_batches(engine, mapping, trial_lanes):
    engine.set_mapping(mapping, trial_lanes=trial_lanes)
    try:
        for c in CASES:
            param = make_parameter(c["channels"], c["bits"], c["max_block_size"], 48000, c["ms"], c["trials"])
            image = engine.encode_host([case_pcm(c)], param)[0]
            assert len(image) == c["aad_bytes"] and sha256(image) == c["aad_sha256"], (mapping, trial_lanes, c)
    finally:
        engine.set_mapping("auto", trial_lanes="dual")


def test_golden_images_legacy_api():
    import aad_amd
    codec = aad_amd.LegacyCodec(aad_amd.load_library())
    for c in CASES[::3]:
        image = codec.encode(case_pcm(c), c["bits"], c["max_block_size"], 48000, c["ms"], c["trials"])
        assert sha256(image) == c["aad_sha256"], c
        assert sha256(codec.decode(image)[0].astype("<i2").tobytes()) == c["decoded_sha256"], c


@pytest.mark.parametrize("trial_lanes", ["dual", "single"])
@pytest.mark.parametrize("trials", [4, 7, 255])
def test_batches_with_high_counts_and_tiles(engine, trials, trial_lanes):
    """ragged same-parameter batches (many streams per launch: the dual layout's slots are per stream), cut into tiles as well"""
    rng = np.random.default_rng(600 + trials)
    engine.set_mapping("auto", trial_lanes=trial_lanes)
    try:
        for ch, bits, ms, mbs in ((2, 4, False, 1024), (2, 3, True, 256), (1, 2, False, 200), (1, 4, False, 128)):
            _, _, spb = ob.geometry(mbs, ch, bits)
            top = 3 * spb + 20 if trials < 100 else spb + 30
            pcms = [synth_pcm(1, int(rng.integers(1, top)), ch, seed=int(rng.integers(0, 1 << 30)), kind=str(rng.choice(["music", "noise"])))[0]
                    for _ in range(40 if trials < 100 else 12)]
            want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
            param = make_parameter(ch, bits, mbs, 48000, ms, trials)
            for tile in (0, 2):
                engine.set_tile_kbytes(tile)
                got = engine.encode_host(pcms, param)
                for i, (g, w) in enumerate(zip(got, want)):
                    assert g == w, (trials, trial_lanes, ch, bits, ms, mbs, tile, i, len(pcms[i]))
    finally:
        engine.set_mapping("auto", trial_lanes="dual")
        engine.set_tile_kbytes(0)


@pytest.mark.parametrize("streams,ch,bits,trials,ms", [
    (2048, 2, 4, 2, False),   # 4096 recurrences: dual layout, two-wave workgroups, one per CU
    (2100, 2, 4, 2, False),   # 4200: dual layout, one-wave workgroups
    (2560, 2, 3, 1, True),    # 5120: the last batch of the dual layout
    (2561, 2, 4, 2, False),   # 5122: the one-after-the-other layout on the quad mapping
    (9000, 1, 2, 1, False),   # 9000 mono recurrences, quad mapping, search then encode
    (8200, 2, 4, 3, False),   # 16 400: the dense trial-search encoder, one-wave workgroups
    (57000, 1, 4, 1, False),  # 57 000 mono lanes: the dense trial-search encoder in four-wave workgroups (its LDS fits three one-wave ones per CU)
    (33000, 2, 3, 1, False),  # 66 000 lanes: beyond one wave per SIMD
])
def test_trial_search_launch_ranges(engine, streams, ch, bits, trials, ms):
    """Both sides of every launch-geometry switch of the trial search (aad_hip_engine.hip: pick_dual / kDualMaxRecurrences, the dual
    kernel's workgroup size, mapping_limits.encode_quad, dense_encode_workgroup) under the default options: device-resident
    one-block batches, sampled streams against the oracle, and every copy of a base stream identical."""
    import torch
    engine.set_mapping("auto", trial_lanes="dual")
    spb = ob.geometry(1024, ch, bits)[2]
    base = synth_pcm(250, spb, ch, seed=31 * streams + bits)
    pcm = np.concatenate([base] * (-(-streams // 250)))[:streams]
    param = make_parameter(ch, bits, 1024, 48000, ms, trials)
    d_img, size = engine.encode_uniform(torch.from_numpy(np.ascontiguousarray(pcm)).cuda(), param)
    torch.cuda.synchronize()
    img = d_img.cpu().numpy()
    for s in list(range(0, 250, 23)) + [249]:
        want = ob.encode(base[s], bits, 1024, 48000, ms, trials)
        for copy in range(s, streams, 250 * max(1, streams // 250 // 3)):
            assert bytes(img[copy, :size]) == want, (streams, ch, bits, trials, s, copy)
    assert np.array_equal(img[:250], img[250:500])
    tail = streams - streams % 250 - 250
    assert np.array_equal(img[tail:tail + 250], img[:250])
