"""GPU parity tests proper: the HIP engine, called through the C-ABI, against
 (a) the reference's own fixtures and the golden hashes made from the compiled reference, and
 (b) the CPU oracle on the same seeded inputs.
Bar: bit-exact (integer/byte work).  Run with -m gpu on an MI355X."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import aad_amd
import oracle_binding as ob
from aad_amd.capi import LANE_STATE_DTYPE, make_parameter
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, ROOT as ROOT_DIR, read_wav16, sha256, wav16_bytes

pytestmark = pytest.mark.gpu
FIX = os.path.join(GOLDEN, "ref_fixtures")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401  (loads the HIP runtime the library then shares)
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(params=["dense", "dense-tiled", "quad", "quad-fused"])
def mapping(request, engine):
    """The lane mappings of the kernels (one lane per recurrence / four lanes per recurrence; for the
    quad decoder both the two-kernel form and the single fused kernel): the host picks by batch
    size, a context option forces one (AADHip_ContextSetOption).  The environment variable is the
    option's default for contexts created from here on - the legacy API's handles."""
    engine.set_mapping(request.param)
    os.environ["AAD_HIP_MAPPING"] = request.param
    yield request.param
    os.environ.pop("AAD_HIP_MAPPING", None)
    engine.set_mapping("auto")


@pytest.fixture(scope="module")
def legacy():
    import torch  # noqa: F401
    return aad_amd.LegacyCodec(aad_amd.load_library())


# ---- (a) reference fixtures / golden hashes -----------------------------------------------

@pytest.mark.parametrize("name", ["sin300Hz_mono", "sin300Hz"])
def test_legacy_api_reproduces_reference_fixtures(legacy, name, mapping):
    """BASELINE config 1 through the GPU path: CLI defaults (4-bit, 1024, trials 2)."""
    pcm, rate = read_wav16(os.path.join(FIX, name + ".wav"))
    gold = open(os.path.join(FIX, name + ".aad"), "rb").read()
    assert legacy.encode(pcm, 4, 1024, rate, False, 2) == gold
    dec, hd = legacy.decode(gold)
    assert wav16_bytes(dec, hd.sampling_rate) == open(os.path.join(FIX, name + "_decoded.wav"), "rb").read()


def test_manifest_cases_batched(engine, mapping):
    """Every golden case, grouped by parameter set into ragged batches (different lengths per stream)."""
    groups = {}
    for c in MANIFEST["cases"]:
        groups.setdefault((c["channels"], c["bits"], c["ms"], c["trials"], c["max_block_size"]), []).append(c)
    for (ch, bits, ms, trials, mbs), cases in groups.items():
        pcms = [synth_pcm(1, c["samples"], ch, seed=c["seed"], kind=c["kind"])[0] for c in cases]
        param = make_parameter(ch, bits, mbs, 48000, ms, trials)
        images = engine.encode_host(pcms, param)
        for c, img in zip(cases, images):
            assert len(img) == c["aad_bytes"] and sha256(img) == c["aad_sha256"], c["name"]
        decoded = engine.decode_host(images)
        for c, d in zip(cases, decoded):
            assert sha256(d.tobytes()) == c["decoded_sha256"], c["name"]


@pytest.mark.parametrize("lanes", ["dual", "single"])
def test_trial_search_lane_layouts(engine, lanes):
    """The trial search on the quad mapping with its probe strand on lanes of its own ("dual", the
    host's choice for streams of three blocks and more) and on the chain's lanes ("single"): every
    golden case with trials, single- and many-block streams alike, under both."""
    engine.set_mapping("quad", lanes)
    try:
        groups = {}
        for c in MANIFEST["cases"]:
            if c["trials"]:
                groups.setdefault((c["channels"], c["bits"], c["ms"], c["trials"], c["max_block_size"]), []).append(c)
        assert len(groups) >= 30
        for (ch, bits, ms, trials, mbs), cases in groups.items():
            pcms = [synth_pcm(1, c["samples"], ch, seed=c["seed"], kind=c["kind"])[0] for c in cases]
            images = engine.encode_host(pcms, make_parameter(ch, bits, mbs, 48000, ms, trials))
            for c, img in zip(cases, images):
                assert sha256(img) == c["aad_sha256"], (lanes, c["name"])
    finally:
        engine.set_mapping("auto", "dual")


@pytest.mark.parametrize("corpus", MANIFEST["corpora"], ids=lambda c: c["name"])
def test_baseline_corpora_device_resident(engine, corpus, mapping):
    """BASELINE configs 2/3/4(stereo)/5 shapes: device-resident uniform batches, hashed against the reference."""
    import torch
    pcm = synth_pcm(corpus["streams"], corpus["samples"], corpus["channels"], seed=corpus["seed"])
    assert sha256(pcm.tobytes()) == corpus["pcm_sha256"]
    param = make_parameter(corpus["channels"], corpus["bits"], corpus["max_block_size"], 48000, False, corpus["trials"])
    d_pcm = torch.from_numpy(pcm).cuda()
    d_img, size = engine.encode_uniform(d_pcm, param)
    torch.cuda.synchronize()
    assert size == corpus["image_bytes"]
    img = d_img.cpu().numpy()
    assert sha256(np.ascontiguousarray(img[:, :size]).tobytes()) == corpus["aad_concat_sha256"]
    assert not img[:, size:].any()
    d_dec, hd = engine.decode_uniform(d_img, size)
    torch.cuda.synchronize()
    assert sha256(d_dec.cpu().numpy().tobytes()) == corpus["decoded_concat_sha256"]


@pytest.mark.parametrize("lanes", ["dual", "single"])
def test_long_chains_with_trial_search_both_lane_layouts(engine, lanes):
    """Long in-lane block chains WITH the trial search (1 stream x 1000 blocks, 40 x 25, mono 2-/3-bit
    multi-block) on the quad mapping under both trial-lane layouts, against the reference's hashes
    (src/aad_encoder.c:853-886 block chain, :470-562 search, :645-653 header carry)."""
    import torch
    engine.set_mapping("quad", lanes)
    try:
        corpora = [c for c in MANIFEST["corpora"] if c["trials"]]
        assert len(corpora) >= 4
        for corpus in corpora:
            pcm = synth_pcm(corpus["streams"], corpus["samples"], corpus["channels"], seed=corpus["seed"])
            param = make_parameter(corpus["channels"], corpus["bits"], 1024, 48000, False, corpus["trials"])
            d_img, size = engine.encode_uniform(torch.from_numpy(pcm).cuda(), param)
            torch.cuda.synchronize()
            img = d_img.cpu().numpy()
            assert sha256(np.ascontiguousarray(img[:, :size]).tobytes()) == corpus["aad_concat_sha256"], (lanes, corpus["name"])
    finally:
        engine.set_mapping("auto", "dual")


@pytest.mark.parametrize("corpus", MANIFEST["eight_channel_corpora"], ids=lambda c: c["name"])
def test_config4_eight_channel_corpora(engine, corpus):
    """BASELINE config 4 at full size: 10 000 eight-channel one-block segments (3- and 2-bit; 4-bit
    at 1000), every (segment, channel) hashed as the mono image the compiled reference produced
    for that channel (tests/golden/make_golden.py), and the decode checked against the oracle."""
    import hashlib
    import torch
    from aad_amd.reframe import channels_as_mono_images
    pcm = synth_pcm(corpus["streams"], corpus["samples"], 8, seed=corpus["seed"])
    assert sha256(pcm.tobytes()) == corpus["pcm_sha256"]
    d_img, size = engine.encode_uniform(torch.from_numpy(pcm).cuda(), make_parameter(8, corpus["bits"], 1024))
    d_dec, _ = engine.decode_uniform(d_img, size)
    torch.cuda.synchronize()
    img = np.ascontiguousarray(d_img.cpu().numpy()[:, :size])
    mono = channels_as_mono_images(img, 8, corpus["bits"], corpus["block_size"], corpus["mono_block_size"])
    assert hashlib.sha256(np.ascontiguousarray(mono).tobytes()).hexdigest() == corpus["mono_images_concat_sha256"]
    dec = d_dec.cpu().numpy()
    for s in range(0, corpus["streams"], 501):
        assert np.array_equal(dec[s], ob.decode(bytes(img[s]))[0]), s


def test_eight_channel_lanes_equal_reference_mono(engine):
    from test_oracle_golden import extract_channel_as_mono
    for bits in (4, 3, 2):
        pcm = synth_pcm(4, 1000, 8, seed=77)
        images = engine.encode_host([pcm[s] for s in range(4)], make_parameter(8, bits, 1024))
        for s in range(4):
            for c in range(8):
                want = [e for e in MANIFEST["eight_channel_as_mono"]
                        if e["bits"] == bits and e["stream"] == s and e["channel"] == c][0]
                assert sha256(extract_channel_as_mono(images[s], c, 128)) == want["aad_sha256"]


# ---- (b) oracle on seeded inputs ------------------------------------------------------------

@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("channels", [1, 2, 3, 8])
def test_ragged_batch_vs_oracle(engine, bits, channels, mapping):
    rng = np.random.default_rng(bits * 100 + channels)
    for trials, ms, mbs in ((0, False, 1024), (2, False, 256), (1, channels == 2, 1024), (0, channels == 2, 18 * channels + 24)):
        lens = [1, 2, 3, 4, 5, 6, 7, 12] + [int(v) for v in rng.integers(8, 5000, 40)]
        kinds = ["music", "noise", "nyquist"]
        pcms = [synth_pcm(1, n, channels, seed=900 + i, kind=kinds[i % 3])[0] for i, n in enumerate(lens)]
        images = engine.encode_host(pcms, make_parameter(channels, bits, mbs, 48000, ms, trials))
        for i, (p, img) in enumerate(zip(pcms, images)):
            assert img == ob.encode(p, bits, mbs, 48000, ms, trials), (i, lens[i], trials, ms, mbs)
        decoded = engine.decode_host(images)
        for i, (img, d) in enumerate(zip(images, decoded)):
            assert np.array_equal(d, ob.decode(img)[0]), (i, lens[i])


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("channels", [1, 2])
def test_dual_trial_search_encodes_beside_the_chain(engine, bits, channels):
    """The dual trial search (aad_encode.hip.h encode_block_dual) encodes every candidate while the chain
    still measures and moves the winners' bytes from scratch slots into the image, channel by channel:
    1-3 trials, first / later / short last blocks, block sizes whose bodies are not multiples of the
    12-byte pieces, M/S, against the oracle.  Both trial-lane layouts must give the same bytes."""
    rng = np.random.default_rng(4242 + bits * 10 + channels)
    kinds = ["music", "noise", "nyquist"]
    for trials in (1, 2, 3):
        for ms, mbs in ((False, 18 * channels + 29), (channels == 2, 200), (False, 1024)):
            lens = [3, 4, 5, 17, 64] + [int(v) for v in rng.integers(20, 6000, 27)]
            pcms = [synth_pcm(1, n, channels, seed=7000 + i, kind=kinds[i % 3])[0] for i, n in enumerate(lens)]
            param = make_parameter(channels, bits, mbs, 48000, ms, trials)
            want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
            try:
                for layout in ("dual", "single"):
                    engine.set_mapping("quad", trial_lanes=layout)
                    images = engine.encode_host(pcms, param)
                    for i, img in enumerate(images):
                        assert img == want[i], (layout, i, lens[i], trials, ms, mbs)
            finally:
                engine.set_mapping("auto", trial_lanes="dual")


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("channels", [1, 2])
def test_dense_kernels_group_boundaries(engine, bits, channels):
    """The dense kernels move code bytes in groups of chunks (decoder: GroupCodes / StereoGroupCodes, encoder:
    CodeStage, paired sample loads, the mono decoder's lead chunk stored with chunk 0): block sizes that put 0 .. 20
    whole chunks and every remainder behind the header, several blocks per stream and a ragged last one, M/S -
    against the oracle, dense mapping forced."""
    rng = np.random.default_rng(9100 + bits * 10 + channels)
    engine.set_mapping("dense")
    try:
        for mbs in list(range(18 * channels + 3, 18 * channels + 3 + 170 * channels, 5 * channels + 1)) + [1024]:
            rc, block_size, spb = ob.geometry(mbs, channels, bits)
            if rc != 0:
                continue
            ms = bool(channels == 2 and mbs % 2)
            lens = [spb, 2 * spb + 1, 3 * spb - 2, 3 * spb + int(rng.integers(5, max(6, spb)))]
            pcms = [synth_pcm(1, n, channels, seed=9200 + i + mbs, kind=["music", "noise"][i % 2])[0] for i, n in enumerate(lens)]
            try:
                want = [ob.encode(p, bits, mbs, 48000, ms, 0) for p in pcms]
            except RuntimeError:  # a block that would carry no data (reference src/aad_encoder.c:170-172)
                continue
            images = engine.encode_host(pcms, make_parameter(channels, bits, mbs, 48000, ms, 0))
            for i, img in enumerate(images):
                assert img == want[i], (mbs, i, lens[i])
            decoded = engine.decode_host(images)
            for i, (img, d) in enumerate(zip(images, decoded)):
                assert np.array_equal(d, ob.decode(img)[0]), (mbs, i, lens[i])
    finally:
        engine.set_mapping("auto")


def test_staging_threads_do_not_change_bytes(engine):
    """Host-memory batches above a megabyte are staged by helper threads (AAD_HIP_OPTION_STAGING_THREADS);
    a ragged 12 MB batch - several chunks' worth at 8 threads' ranges, one 1.2 MB stream among short
    ones - must come back identical whatever the thread count, and equal to the oracle on a sample."""
    rng = np.random.default_rng(77)
    lens = [int(v) for v in rng.integers(1, 9000, 700)]
    lens[13] = 300000
    lens[699] = 1
    kinds = ["music", "noise", "nyquist"]
    pcms = [synth_pcm(1, n, 2, seed=5000 + i, kind=kinds[i % 3])[0] for i, n in enumerate(lens)]
    param = make_parameter(2, 4, 1024, 48000, False, 0)
    try:
        engine.set_staging_threads(1)
        want_images = engine.encode_host(pcms, param)
        want_pcm = engine.decode_host(want_images)
        for threads in (3, 8, 0):
            engine.set_staging_threads(threads)
            images = engine.encode_host(pcms, param)
            assert images == want_images, threads
            decoded = engine.decode_host(images)
            assert all(np.array_equal(a, b) for a, b in zip(decoded, want_pcm)), threads
    finally:
        engine.set_staging_threads(0)
    for i in (0, 13, 350, 698, 699):
        assert want_images[i] == ob.encode(pcms[i], 4, 1024, 48000, False, 0), i
        assert np.array_equal(want_pcm[i], ob.decode(want_images[i])[0]), i
    with pytest.raises(Exception):
        engine.set_staging_threads(9)


@pytest.mark.parametrize("kbytes", [1, 24, 200])
@pytest.mark.parametrize("bits,channels,ms,trials,mbs", [(4, 2, False, 0, 256), (4, 2, True, 1, 128), (3, 1, False, 0, 64),
                                                         (2, 3, False, 2, 96), (4, 8, False, 0, 256)])
def test_tiled_host_path_is_tile_size_independent(engine, kbytes, bits, channels, ms, trials, mbs):
    """The host-memory path cuts a batch into groups of streams and tiles of blocks, the encoder's
    predictor state staying on the device between a group's tiles (AAD_HIP_OPTION_TILE_KBYTES forces
    the cut on small data).  Ragged lengths - one-sample streams next to 100-block ones, streams that
    end inside a tile - must give the oracle's images and PCM whatever the tile size, with and without
    caller-held state, and truncated images must decode like the reference's block walk."""
    rng = np.random.default_rng(kbytes * 7 + bits)
    lens = [1, 3, 4, 5, 9000, 2, 700, 12000] + [int(v) for v in rng.integers(1, 4000, 22)]
    kinds = ["music", "noise", "nyquist"]
    pcms = [synth_pcm(1, n, channels, seed=300 + i, kind=kinds[i % 3])[0] for i, n in enumerate(lens)]
    param = make_parameter(channels, bits, mbs, 48000, ms, trials)
    want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
    try:
        engine.set_tile_kbytes(kbytes)
        images = engine.encode_host(pcms, param)
        for i, (img, w) in enumerate(zip(images, want)):
            assert img == w, (i, lens[i])
        decoded = engine.decode_host(images)
        for i, d in enumerate(decoded):
            assert np.array_equal(d, ob.decode(want[i])[0]), (i, lens[i])
        # caller-held state across two calls (the reference's reused handle, src/aad_encoder.c:853-886)
        state = np.zeros(len(pcms) * channels, dtype=LANE_STATE_DTYPE)
        lanes = [ob.fresh_lanes(channels) for _ in pcms]
        for k in range(2):
            state["stepsize_index"] = 0
            images = engine.encode_host(pcms, param, state=state)
            for i, p in enumerate(pcms):
                assert images[i] == ob.encode(p, bits, mbs, 48000, ms, trials, lanes=lanes[i], reset_idx=True), (k, i)
                for c in range(channels):
                    assert list(state[i * channels + c]["weight"]) == list(lanes[i][c].w), (k, i, c)
                    assert int(state[i * channels + c]["stepsize_index"]) == lanes[i][c].idx
        # images that end early: inside a block header is an error, anywhere else the walk stops
        from aad_amd.engine import parse_header
        head, bs = 18 * channels, parse_header(want[0][:31]).block_size

        def cut_at(w, drop):
            n = max(31 + head, len(w) - drop)
            inside = (n - 31) % bs
            if 0 < inside < head:
                n += head - inside
            return w[: min(n, len(w))]
        cut = [cut_at(w, 37 * (i % 5)) for i, w in enumerate(want)]
        engine.set_tile_kbytes(0)
        whole = engine.decode_host(cut)
        engine.set_tile_kbytes(kbytes)
        tiled = engine.decode_host(cut)
        for i, (x, y) in enumerate(zip(tiled, whole)):
            assert np.array_equal(x, y), (i, lens[i])
    finally:
        engine.set_tile_kbytes(0)


def test_state_carry_matches_oracle(engine, mapping):
    """encoder state in/out == the reference's reused-handle behaviour (src/aad_encoder.c:853-886)"""
    ch, streams = 2, 5
    state = np.zeros(streams * ch, dtype=LANE_STATE_DTYPE)
    lanes = [ob.fresh_lanes(ch) for _ in range(streams)]
    for k in range(3):
        pcms = [synth_pcm(1, 1500 + 37 * s + k, ch, seed=10 * k + s)[0] for s in range(streams)]
        state["stepsize_index"] = 0   # what SetEncodeParameter does between calls
        images = engine.encode_host(pcms, make_parameter(ch, 4, 1024, 48000, False, 1), state=state)
        for s in range(streams):
            assert images[s] == ob.encode(pcms[s], 4, 1024, 48000, False, 1, lanes=lanes[s], reset_idx=True), (k, s)
            for c in range(ch):
                assert list(state[s * ch + c]["weight"]) == list(lanes[s][c].w)
                assert int(state[s * ch + c]["stepsize_index"]) == lanes[s][c].idx


def test_legacy_handle_reuse_and_decode_block(legacy):
    lib = legacy.lib
    enc = lib.AADEncoder_Create(1024, None, 0)
    lanes = ob.fresh_lanes(2)
    try:
        for k in range(2):
            pcm = synth_pcm(1, 2100 + k, 2, seed=70 + k)[0]
            img = legacy.encode(pcm, 4, 1024, 48000, True, 2, encoder=enc)
            assert img == ob.encode(pcm, 4, 1024, 48000, True, 2, lanes=lanes, reset_idx=True)
    finally:
        lib.AADEncoder_Destroy(enc)
    hd = legacy.decode_header(img)
    full = ob.decode(img)[0]
    spb, bs = hd.num_samples_per_block, hd.block_size
    for b in range(3):
        blk = img[31 + b * bs: 31 + (b + 1) * bs]
        want = full[b * spb:(b + 1) * spb]
        got = legacy.decode_block(hd, blk, min(spb, len(want)))
        assert np.array_equal(got, want), b
    assert np.array_equal(legacy.decode_block(hd, img[31:31 + bs], 10), full[:10])   # short buffer: decode until full


def test_truncated_image_decodes_like_block_walk(engine, mapping):
    pcm = synth_pcm(1, 5000, 2, seed=5)[0]
    img = ob.encode(pcm, 4, 1024)
    cut = img[: 31 + 1024 * 2 + 500]   # third block cut short: missing bytes read as zero
    want = np.zeros((5000, 2), dtype=np.int16)
    buf = np.frombuffer(cut, dtype=np.uint8)
    ob.lib().aado_decode_stream(buf.ctypes.data, len(buf), 8, want.ctypes.data, 5000, None)
    got = engine.decode_host([cut])[0]
    assert np.array_equal(got[: 3 * 992], want[: 3 * 992])


def test_fp64_rmse_selection_corner(engine):
    """int32-wrapped squares / NaN compare corner (SURVEY.md finding 5): full-scale noise and a
    Nyquist square drive |qd| past 46341 with trials on."""
    for kind in ("noise", "nyquist"):
        for bits in (4, 2):
            pcms = [synth_pcm(1, 4000, 1, seed=s, kind=kind)[0] for s in range(8)]
            images = engine.encode_host(pcms, make_parameter(1, bits, 1024, 48000, False, 2))
            for p, img in zip(pcms, images):
                assert img == ob.encode(p, bits, 1024, 48000, False, 2)


def test_plan_validation_errors(engine):
    from aad_amd import AADApiResult as R, ApiError
    from aad_amd.capi import STREAM_DESC_DTYPE
    d = np.zeros(1, dtype=STREAM_DESC_DTYPE)
    d["num_samples"], d["data_size"] = 992, 10
    with pytest.raises(ApiError) as e:
        engine.encode_plan(make_parameter(2, 4, 1024), d)
    assert e.value.code == R.INSUFFICIENT_BUFFER
    d["data_size"] = 4096
    for bad in (make_parameter(9, 4), make_parameter(2, 1), make_parameter(2, 5), make_parameter(1, 4, ms=True),
                make_parameter(3, 4, ms=True), make_parameter(2, 4, 20)):
        with pytest.raises(ApiError) as e:
            engine.encode_plan(bad, d)
        assert e.value.code == R.INVALID_FORMAT
    d["num_samples"] = 0
    with pytest.raises(ApiError) as e:
        engine.encode_plan(make_parameter(2, 4, 1024), d)
    assert e.value.code == R.INVALID_FORMAT


@pytest.mark.parametrize("kbytes", [0, 4])
def test_host_batch_errors_leave_nothing_behind(engine, kbytes):
    """One bad stream fails the whole host-memory call before any tile runs - also when the batch is
    cut into tiles: an image that ends inside a block header is the reference's INSUFFICIENT_DATA
    (src/aad_decoder.c:347-349), an output buffer that is too small INSUFFICIENT_BUFFER, and the other
    streams' outputs stay untouched.  The context keeps working afterwards."""
    from aad_amd import AADApiResult as R
    pcms = [synth_pcm(1, 3000 + 100 * i, 2, seed=40 + i)[0] for i in range(6)]
    param = make_parameter(2, 4, 256, 48000, False, 0)
    images = [ob.encode(p, 4, 256) for p in pcms]
    lib, ctx = engine.lib, engine._ctx
    try:
        engine.set_tile_kbytes(kbytes)
        n = len(images)
        bad = list(images)
        bad[4] = bad[4][: 31 + 256 * 3 + 20]                     # inside block 3's 36-byte header
        bufs = [np.frombuffer(b, dtype=np.uint8) for b in bad]
        outs = [np.full((p.shape[0], 2), 77, dtype=np.int16) for p in pcms]
        sizes = np.array([len(b) for b in bufs], dtype=np.uint64)
        caps = np.array([p.shape[0] for p in pcms], dtype=np.uint32)
        got = np.full(n, 99, dtype=np.uint32)
        dp = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        pp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        assert lib.AADHip_DecodeBatch(ctx, n, dp, sizes.ctypes.data, pp, caps.ctypes.data, got.ctypes.data) == R.INSUFFICIENT_DATA
        assert all((o == 77).all() for o in outs)
        # encode: stream 2's buffer one byte short
        pcm_c = [np.ascontiguousarray(p) for p in pcms]
        nsamp = np.array([p.shape[0] for p in pcm_c], dtype=np.uint32)
        ecaps = np.array([len(b) for b in images], dtype=np.uint64)
        ecaps[2] -= 1
        eouts = [np.full(len(b), 0x55, dtype=np.uint8) for b in images]
        esizes = np.zeros(n, dtype=np.uint64)
        ip = (C.c_void_p * n)(*[p.ctypes.data for p in pcm_c])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in eouts])
        assert lib.AADHip_EncodeBatch(ctx, C.byref(param), n, ip, nsamp.ctypes.data, op, ecaps.ctypes.data,
                                      esizes.ctypes.data, None) == R.INSUFFICIENT_BUFFER
        assert all((o == 0x55).all() for o in eouts)
        # and the same context still produces the oracle's bytes
        assert engine.encode_host(pcms, param) == images
    finally:
        engine.set_tile_kbytes(0)


def test_reference_cli_linked_against_this_library(tmp_path):
    """the reference's own main.c / wav.c / option parser, compiled from
    its sources in the build container and linked against libaad_hip.so instead of the reference
    codec objects (oracle/_ref/aad_on_hip), reproduces the reference's fixtures with its default
    options - `aad -e` / `aad -d` as in reference test/make_test_data.sh:4-7."""
    import subprocess
    cli = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "aad_on_hip")
    if not os.path.exists(cli):
        pytest.skip("oracle/_ref/aad_on_hip not built")
    for name in ("sin300Hz_mono", "sin300Hz"):
        out = tmp_path / (name + ".aad")
        subprocess.run([cli, "-e", os.path.join(FIX, name + ".wav"), str(out)], check=True,
                       stdout=subprocess.DEVNULL)
        assert out.read_bytes() == open(os.path.join(FIX, name + ".aad"), "rb").read()
        wav = tmp_path / (name + "_decoded.wav")
        subprocess.run([cli, "-d", str(out), str(wav)], check=True, stdout=subprocess.DEVNULL)
        assert wav.read_bytes() == open(os.path.join(FIX, name + "_decoded.wav"), "rb").read()


# ---- BASELINE full sizes: size-independent properties + sampled oracle checks ------------------

def _hash_rows(arr):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(arr).tobytes())
    return h.hexdigest()


def test_config5_full_size_10000_files(engine):
    """BASELINE config 5 shape at full size on one GPU: 10 000 stereo 4-bit files x 10 blocks.
    Properties: image size = format arithmetic; the two lane mappings (very different code paths)
    give identical bytes; every 97th stream is bit-exact against the oracle; decode(encode(x))
    equals the oracle's decode of the same image."""
    import torch
    streams, samples = 10000, 9920
    pcm = synth_pcm(streams, samples, 2, seed=1234)
    param = make_parameter(2, 4, 1024)
    d_pcm = torch.from_numpy(pcm).cuda()
    results = {}
    for mode in ("dense", "quad"):
        engine.set_mapping(mode)
        try:
            d_img, size = engine.encode_uniform(d_pcm, param)
            d_dec, hd = engine.decode_uniform(d_img, size)
            torch.cuda.synchronize()
        finally:
            engine.set_mapping("auto")
        results[mode] = (_hash_rows(d_img.cpu().numpy()), _hash_rows(d_dec.cpu().numpy()))
        assert size == 31 + 10 * 1024
    assert results["dense"] == results["quad"]
    img, dec = d_img.cpu().numpy(), d_dec.cpu().numpy()
    for s in range(0, streams, 97):
        want = ob.encode(pcm[s], 4, 1024)
        assert bytes(img[s, :size]) == want, s
        assert np.array_equal(dec[s], ob.decode(want)[0]), s
    # lossy codec, but a sane one: reconstruction error far below the signal (reference
    # test/test_aad_encode_decode.c:303-420 uses RMSE thresholds of the same kind)
    err = (dec.astype(np.float64) - pcm.astype(np.float64)) / 32768.0
    assert float(np.sqrt(np.mean(err ** 2))) < 0.05


@pytest.mark.parametrize("bits", [3, 2])
def test_config4_full_size_8_channels(engine, bits):
    """BASELINE config 4: 10 000 eight-channel one-block segments, 3-bit and 2-bit (the container
    extension; pinned per channel against the reference's mono encodes in
    test_eight_channel_lanes_equal_reference_mono)."""
    import torch
    spb = {3: 292, 2: 444}[bits]
    streams = 10000
    pcm = synth_pcm(streams, spb, 8, seed=4321)
    param = make_parameter(8, bits, 1024)
    d_img, size = engine.encode_uniform(torch.from_numpy(pcm).cuda(), param)
    d_dec, hd = engine.decode_uniform(d_img, size)
    torch.cuda.synchronize()
    assert size == 31 + {3: 1008, 2: 1024}[bits] and hd.num_samples_per_block == spb
    img, dec = d_img.cpu().numpy(), d_dec.cpu().numpy()
    for s in range(0, streams, 211):
        want = ob.encode(pcm[s], bits, 1024)
        assert bytes(img[s, :size]) == want, s
        assert np.array_equal(dec[s], ob.decode(want)[0]), s


def test_aad_batch_cli_matches_reference_cli_outputs(tmp_path):
    """N1/N2: the many-file C front end (aad_amd/cli/aad_batch.c) with the reference's default
    options writes the same bytes as `aad -e` / `aad -d` (the fixtures of test/make_test_data.sh),
    mono and stereo inputs mixed in one invocation."""
    import subprocess
    cli = os.path.join(os.path.dirname(GOLDEN), "..", "aad_amd", "aad_batch")
    assert os.path.exists(cli), "aad_batch not built"
    enc_dir, dec_dir = tmp_path / "enc", tmp_path / "dec"
    enc_dir.mkdir()
    dec_dir.mkdir()
    names = ("sin300Hz_mono", "sin300Hz")
    subprocess.run([cli, "-e", "-o", str(enc_dir)] + [os.path.join(FIX, n + ".wav") for n in names], check=True)
    for n in names:
        assert (enc_dir / (n + ".aad")).read_bytes() == open(os.path.join(FIX, n + ".aad"), "rb").read()
    subprocess.run([cli, "-d", "-o", str(dec_dir)] + [str(enc_dir / (n + ".aad")) for n in names], check=True)
    for n in names:
        assert (dec_dir / (n + ".wav")).read_bytes() == open(os.path.join(FIX, n + "_decoded.wav"), "rb").read()
    # other option values against the oracle
    subprocess.run([cli, "-e", "-b", "3", "-s", "256", "-t", "0", "-m", "-o", str(enc_dir),
                    os.path.join(FIX, "unit_impulse.wav")], check=True)
    pcm, rate = read_wav16(os.path.join(FIX, "unit_impulse.wav"))
    assert (enc_dir / "unit_impulse.aad").read_bytes() == ob.encode(pcm, 3, 256, rate, True, 0)


def test_random_access_block_ranges(engine):
    """every block header carries the full predictor state, so any block
    range decodes on its own - here as a batch of bare blocks (has_file_header = 0) picked out of
    the middle of several streams, equal to the same slices of the full decodes."""
    import torch
    from aad_amd.capi import STREAM_DESC_DTYPE
    from aad_amd.engine import parse_header
    pcms = [synth_pcm(1, 9000 + 500 * i, 2, seed=300 + i)[0] for i in range(4)]
    images = [ob.encode(p, 4, 1024, 48000, False, 1) for p in pcms]
    full = [ob.decode(img)[0] for img in images]
    hd = parse_header(images[0][:31])
    spb, bs = hd.num_samples_per_block, hd.block_size
    picks = [(0, 3, 2), (1, 0, 1), (2, 5, 4), (3, 8, 1), (0, 9, 1)]   # (stream, first block, block count)
    blob, descs, pcm_off = bytearray(), [], 0
    for s, b0, nb in picks:
        chunk = images[s][31 + b0 * bs: 31 + (b0 + nb) * bs]
        frames = min(nb * spb, len(full[s]) - b0 * spb)
        descs.append((pcm_off, len(blob), len(chunk), frames, 0))
        blob += chunk + bytes(-len(chunk) % 16)
        pcm_off += frames * 2
    d = np.array(descs, dtype=STREAM_DESC_DTYPE)
    plan = engine.decode_plan(hd, d, has_file_header=False)
    d_data = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    d_pcm = torch.zeros(pcm_off, dtype=torch.int16, device="cuda")
    plan.run(d_data, d_pcm)
    torch.cuda.synchronize()
    got = d_pcm.cpu().numpy()
    for (s, b0, nb), row in zip(picks, descs):
        frames = row[3]
        assert np.array_equal(got[row[0]: row[0] + frames * 2].reshape(-1, 2), full[s][b0 * spb: b0 * spb + frames]), (s, b0)


def test_extreme_block_sizes(engine, mapping):
    """geometry extremes: the largest block the 16-bit header field allows and the smallest that
    still carries data (reference src/aad_encoder.c:85-131)"""
    for ch, bits, mbs, n in ((2, 4, 65535, 70000), (1, 3, 65535, 180000), (2, 4, 38, 500), (1, 2, 19, 300), (2, 3, 42, 777)):
        pcm = synth_pcm(2, n, ch, seed=mbs)
        images = engine.encode_host([pcm[0], pcm[1]], make_parameter(ch, bits, mbs, 48000, False, 0))
        for s in range(2):
            assert images[s] == ob.encode(pcm[s], bits, mbs), (ch, bits, mbs)
        dec = engine.decode_host(images)
        for s in range(2):
            assert np.array_equal(dec[s], ob.decode(images[s])[0]), (ch, bits, mbs)


@pytest.mark.parametrize("streams,bits,ch", [(3000, 4, 2), (7000, 4, 1), (4096, 4, 2), (4097, 4, 2), (8192, 4, 2), (8193, 4, 2),
                                             (2048, 3, 2), (2049, 3, 2), (4096, 3, 2), (4097, 3, 2), (9000, 3, 2), (4096, 2, 2), (4097, 2, 2),
                                             (4096, 2, 1), (8192, 2, 1), (8193, 4, 1), (12289, 4, 1), (16384, 3, 1), (17000, 3, 1)])
def test_decode_mapping_ranges_auto(engine, streams, bits, ch):
    """The host's own choice of mapping on both sides of every threshold of its per-geometry table
    (mapping option "auto"; aad_hip_engine.hip mapping_limits: encode quad up to 16384 recurrences,
    decode split up to 8192 recurrences - 4096 of them with the residual rows in LDS -, dense beyond): split
    decoder with the residuals in LDS / in a device scratch buffer, dense kernel -
    one-block streams, sampled against the oracle, and the whole batch through the round trip
    decode(encode(x)) == oracle decode."""
    import torch
    engine.set_mapping("auto")
    spb = ob.geometry(1024, ch, bits)[2]
    base = synth_pcm(500, spb, ch, seed=2024 + streams)
    pcm = np.concatenate([base] * (-(-streams // 500)))[:streams]
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    param = make_parameter(ch, bits, 1024, 48000, False, 0)
    d_img, size = engine.encode_uniform(d_pcm, param)
    d_dec, _ = engine.decode_uniform(d_img, size)
    torch.cuda.synchronize()
    img, dec = d_img.cpu().numpy(), d_dec.cpu().numpy()
    want = {}
    for s in list(range(0, 500, 61)) + [499]:
        w_img = ob.encode(base[s], bits, 1024)
        want[s] = (w_img, ob.decode(w_img)[0])
    for s in range(streams):
        if s % 500 in want:
            w_img, w_dec = want[s % 500]
            assert bytes(img[s, :size]) == w_img, s
            assert np.array_equal(dec[s], w_dec), s
    # every copy of a base stream decodes identically
    assert np.array_equal(dec[:500], dec[500:1000])
    assert np.array_equal(dec[streams - 500:], dec[(streams - 500) % 500:][:500]) if streams % 500 == 0 else True


def test_inconsistent_header_geometry(engine, mapping):
    """A header whose samples-per-block does not match its block size (nothing in the decoder's
    checks forbids it, reference src/aad_decoder.c:173-225): one huge block claimed, far more
    samples than the bytes can hold.  Bytes past the image read as zero here and in the oracle;
    the point is that every decode path sizes its buffers from 64-bit arithmetic and survives."""
    pcm = synth_pcm(1, 992, 2, seed=5)[0]
    good = bytearray(ob.encode(pcm, 4, 1024))
    for spb, n in ((1_000_000, 992), (1_000_000, 5000), (0xFFFFFFF0, 3000), (20, 992)):
        img = bytearray(good)
        img[14:18] = n.to_bytes(4, "big")          # num_samples
        img[26:30] = spb.to_bytes(4, "big")        # num_samples_per_block
        try:
            want = ob.decode(bytes(img))[0]
        except RuntimeError:
            with pytest.raises(Exception):
                engine.decode_host([bytes(img)])
            continue
        got = engine.decode_host([bytes(img)])[0]
        assert np.array_equal(got, want), (spb, n)


def test_encode_decode_pipeline_keeps_steps_apart(engine):
    """EncodeDecodePipeline overlaps the encode of step k+1 with the decode of step k on two contexts
    (what bench.py times).  Forty steps - five times round its ring of image buffers - with a different
    batch every step: every step's images and decoded PCM must be the oracle's for THAT step's input,
    whatever ran concurrently."""
    import torch
    from aad_amd.engine import Engine, EncodeDecodePipeline
    second = Engine(0, stream=torch.cuda.Stream(0))
    try:
        streams, samples, steps = 48, 1500, 40
        param = make_parameter(2, 4, 1024, 48000, False, 0)
        pipe = EncodeDecodePipeline(engine, second, param, streams, samples)
        batches = [synth_pcm(streams, samples, 2, seed=7000 + i, kind=("music", "noise", "nyquist")[i % 3]) for i in range(5)]
        d_in = [torch.from_numpy(b).cuda() for b in batches]
        d_out = [torch.zeros_like(d_in[0]) for _ in range(steps)]
        for k in range(steps):
            pipe.step(d_in[k % 5], d_out[k])
        torch.cuda.synchronize()
        want_img = [[ob.encode(b[s], 4, 1024) for s in range(streams)] for b in batches]
        want_pcm = [[ob.decode(w)[0] for w in ws] for ws in want_img]
        for k in range(steps):
            got = d_out[k].cpu().numpy()
            for s in range(0, streams, 5):
                assert np.array_equal(got[s], want_pcm[k % 5][s]), (k, s)
        for k in range(steps - 8, steps):  # the ring still holds the images of the last eight steps
            img = pipe.images[k % 8].cpu().numpy()
            for s in range(0, streams, 7):
                assert bytes(img[s, :pipe.enc.image_size]) == want_img[k % 5][s], (k, s)
        pipe.close()
    finally:
        second.close()


def test_legacy_whole_file_calls_through_tiles(tmp_path):
    """AADEncoder_EncodeWhole / AADDecoder_DecodeWhole on a 120-block stereo file with the host-memory path
    forced into 64 KiB tiles (AAD_HIP_TILE_KBYTES, read when the handle's context is created - hence a
    process of its own): planar int32 rows sliced by block range, the trial search's look-back block
    carried as lead frames, M/S, the handle's state - the image and the PCM must be the oracle's."""
    script = tmp_path / "legacy_tiles.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import torch  # noqa: F401
import aad_amd, oracle_binding as ob
from aad_amd.capi import LegacyCodec
from aad_amd.synth import synth_pcm
legacy = LegacyCodec(aad_amd.load_library())
for ch, bits, ms, trials in ((2, 4, True, 2), (1, 3, False, 1), (2, 2, False, 0)):
    spb = ob.geometry(1024, ch, bits)[2]
    pcm = synth_pcm(1, 120 * spb + 77, ch, seed=31 + bits)[0]
    img = legacy.encode(pcm, bits, 1024, 48000, ms, trials)
    assert img == ob.encode(pcm, bits, 1024, 48000, ms, trials), (ch, bits, ms, trials)
    assert np.array_equal(legacy.decode(img)[0], ob.decode(img)[0]), (ch, bits, ms, trials)
print("ok")
''' % (ROOT_DIR, os.path.join(ROOT_DIR, "tests")))
    env = dict(os.environ, AAD_HIP_TILE_KBYTES="64")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
