"""Pin the oracle (oracle/aad_oracle.c) against the reference's own fixtures and against the
golden vectors generated from the compiled reference (tests/golden/make_golden.py).
CPU only; runs on the GPU box too (needs nothing from /root/reference)."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, cli_mode_input, read_wav16, sha256, wav16_bytes

FIX = os.path.join(GOLDEN, "ref_fixtures")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.mark.parametrize("name", ["sin300Hz_mono", "sin300Hz"])
def test_reference_fixture_encode(name):
    """test/make_test_data.sh:4-7 made the .aad fixtures with the CLI defaults (4-bit, 1024, trials 2)."""
    pcm, rate = read_wav16(os.path.join(FIX, name + ".wav"))
    gold = open(os.path.join(FIX, name + ".aad"), "rb").read()
    assert ob.encode(pcm, bits=4, max_block_size=1024, rate=rate, trials=2) == gold


@pytest.mark.parametrize("name", ["sin300Hz_mono", "sin300Hz"])
def test_reference_fixture_decode(name):
    """reference test/test_aad_decoder.c:256-339 - the suite's only bit-exact pins."""
    gold = open(os.path.join(FIX, name + ".aad"), "rb").read()
    pcm, hd = ob.decode(gold)
    assert wav16_bytes(pcm, hd.sampling_rate) == open(os.path.join(FIX, name + "_decoded.wav"), "rb").read()


def test_geometry_known_answers():
    """reference test/test_aad_encoder.c:33-57 and SURVEY.md section 2b table"""
    kat = {(1024, 1, 4): (1024, 2016), (1024, 2, 4): (1024, 992), (1024, 1, 3): (1023, 2684),
           (1024, 2, 3): (1020, 1316), (1024, 1, 2): (1024, 4028), (1024, 2, 2): (1024, 1980),
           (128, 1, 3): (126, 292), (1024, 8, 4): (1024, 224), (1024, 8, 3): (1008, 292), (1024, 8, 2): (1024, 444)}
    for (mbs, ch, bits), want in kat.items():
        assert ob.geometry(mbs, ch, bits) == (0,) + want
    assert ob.geometry(17, 1, 4)[0] == 2 and ob.geometry(1024, 0, 4)[0] == 2 and ob.geometry(1024, 1, 5)[0] == 2


def test_tables():
    t = np.ctypeslib.as_array(ob.lib().aado_step_table(), (256,))
    assert t[0] == 1 and t[255] == 32767 and int(t.sum()) == 864159 and np.all(np.diff(t.astype(int)) > 0)
    assert list(np.ctypeslib.as_array(ob.lib().aado_index_deltas(4), (8,))) == [-18, -17, -14, 16, 32, 64, 128, 256]
    assert list(np.ctypeslib.as_array(ob.lib().aado_index_deltas(3), (4,))) == [-16, -15, 32, 128]
    assert list(np.ctypeslib.as_array(ob.lib().aado_index_deltas(2), (2,))) == [-14, 40]


@pytest.mark.parametrize("case", MANIFEST["cases"], ids=lambda c: c["name"])
def test_manifest_case(case):
    pcm = synth_pcm(1, case["samples"], case["channels"], seed=case["seed"], kind=case["kind"])[0]
    assert sha256(pcm.tobytes()) == case["pcm_sha256"], "synthetic corpus generator drifted"
    aad = ob.encode(pcm, case["bits"], case["max_block_size"], 48000, case["ms"], case["trials"])
    assert len(aad) == case["aad_bytes"] and sha256(aad) == case["aad_sha256"]
    dec, _ = ob.decode(aad)
    assert sha256(dec.tobytes()) == case["decoded_sha256"]
    if "file" in case:
        assert open(os.path.join(GOLDEN, case["file"]), "rb").read() == aad


def test_eight_channel_equals_mono_lanes():
    """an 8-channel stream's channel c carries the same header fields and
    codes as the reference's mono encode of that channel at the matching block geometry."""
    for bits in (4, 3, 2):
        pcm = synth_pcm(4, 1000, 8, seed=77)
        for s in range(4):
            aad8 = ob.encode(pcm[s], bits, 1024)
            for c in range(8):
                want = [e for e in MANIFEST["eight_channel_as_mono"]
                        if e["bits"] == bits and e["stream"] == s and e["channel"] == c][0]
                assert sha256(extract_channel_as_mono(aad8, c, 128)) == want["aad_sha256"]


@pytest.mark.parametrize("corpus", MANIFEST["corpora"], ids=lambda c: c["name"])
def test_corpus_hashes(corpus):
    """The BASELINE corpora and the long in-lane block chains (1 x 1000 blocks with and without the
    trial search, 1000 x 16 blocks, multi-block mono 2-/3-bit) against the compiled reference's hashes."""
    import hashlib
    pcm = synth_pcm(corpus["streams"], corpus["samples"], corpus["channels"], seed=corpus["seed"])
    assert sha256(pcm.tobytes()) == corpus["pcm_sha256"], "synthetic corpus generator drifted"
    stride = ob.encoded_size(corpus["samples"], corpus["channels"], corpus["bits"], corpus["max_block_size"])
    assert stride == corpus["image_bytes"]
    img = np.zeros((corpus["streams"], stride), dtype=np.uint8)
    assert ob.lib().aado_encode_batch(pcm.ctypes.data, corpus["streams"], corpus["samples"], corpus["channels"], 48000,
                                      corpus["bits"], corpus["max_block_size"], 0, corpus["trials"], img.ctypes.data, stride) == 0
    assert hashlib.sha256(img.tobytes()).hexdigest() == corpus["aad_concat_sha256"]
    dec = np.zeros_like(pcm)
    assert ob.lib().aado_decode_batch(img.ctypes.data, corpus["streams"], stride, stride, dec.ctypes.data, corpus["samples"]) == 0
    assert hashlib.sha256(dec.tobytes()).hexdigest() == corpus["decoded_concat_sha256"]


def test_file_corpus_prefix_hashes_are_consistent():
    """BASELINE config 5's 10 000-file corpus is pinned by prefix hashes (1250 / 2500 / 5000 / 10000 files,
    what 1 / 2 / 4 / 8 ranks x 1250 files gather); its first 1250 files ARE the 1250-file corpus above,
    and the oracle reproduces the 2500-file prefix."""
    import hashlib
    fc = MANIFEST["file_corpora"][0]
    shard = [c for c in MANIFEST["corpora"] if c["name"] == "cfg5_stereo4_1250x10blk_t0"][0]
    assert fc["aad_prefix_sha256"]["1250"] == shard["aad_concat_sha256"]
    pcm = synth_pcm(2500, fc["samples"], 2, seed=fc["seed"])
    img = np.zeros((2500, fc["image_bytes"]), dtype=np.uint8)
    assert ob.lib().aado_encode_batch(pcm.ctypes.data, 2500, fc["samples"], 2, 48000, 4, 1024, 0, 0, img.ctypes.data, fc["image_bytes"]) == 0
    assert hashlib.sha256(img.tobytes()).hexdigest() == fc["aad_prefix_sha256"]["2500"]


@pytest.mark.parametrize("corpus", MANIFEST["eight_channel_corpora"], ids=lambda c: c["name"])
def test_eight_channel_corpus_hashes(corpus):
    """BASELINE config 4 at full size (10 000 eight-channel one-block segments, 3- and 2-bit): every
    (segment, channel) re-framed as a mono image equals the reference's mono encode of that channel."""
    import hashlib
    from aad_amd.reframe import channels_as_mono_images
    pcm = synth_pcm(corpus["streams"], corpus["samples"], 8, seed=corpus["seed"])
    assert sha256(pcm.tobytes()) == corpus["pcm_sha256"]
    stride = ob.encoded_size(corpus["samples"], 8, corpus["bits"], 1024)
    img = np.zeros((corpus["streams"], stride), dtype=np.uint8)
    assert ob.lib().aado_encode_batch(pcm.ctypes.data, corpus["streams"], corpus["samples"], 8, 48000, corpus["bits"], 1024, 0, 0,
                                      img.ctypes.data, stride) == 0
    mono = channels_as_mono_images(img, 8, corpus["bits"], corpus["block_size"], corpus["mono_block_size"])
    assert mono.shape[2] == corpus["mono_image_bytes"]
    assert hashlib.sha256(np.ascontiguousarray(mono).tobytes()).hexdigest() == corpus["mono_images_concat_sha256"]
    # the vectorised re-framing agrees with the byte-by-byte one
    assert bytes(mono[3, 5]) == extract_channel_as_mono(bytes(img[3]), 5, 128)


def extract_channel_as_mono(aad, c, mono_max_block_size):
    """Re-frame channel c of a multi-channel image as the mono image with the same samples/block."""
    import math
    hd = ob.AadoHeader()
    buf = np.frombuffer(aad, dtype=np.uint8)
    assert ob.lib().aado_get_header(buf.ctypes.data, len(buf), hd) == 0
    ch, bits = hd.num_channels, hd.bits_per_sample
    rc, mono_bs, mono_spb = ob.geometry(mono_max_block_size, 1, bits)
    assert rc == 0 and mono_spb == hd.samples_per_block
    ub = math.lcm(8, bits) // 8
    out = bytearray(aad[:31])
    out[12:14] = (1).to_bytes(2, "big")
    out[24:26] = mono_bs.to_bytes(2, "big")
    pos = 31
    while pos < len(aad):
        blk = aad[pos:pos + hd.block_size]
        out += blk[18 * c:18 * (c + 1)]
        body = blk[18 * ch:]
        for u in range(len(body) // (ub * ch)):
            out += body[(u * ch + c) * ub:(u * ch + c + 1) * ub]
        pos += hd.block_size
    return bytes(out)


def _cli_mode_cases():
    with open(os.path.join(GOLDEN, "cli_modes.json")) as f:
        return json.load(f)["cases"]


def test_oracle_reconstruction_modes_match_reference_cli():
    """N3 oracle pin: encode -> decode -> (reconstructed WAV, residual WAV, `-c` statistics line)
    restated in oracle/aad_oracle.c against what the compiled reference CLI wrote / printed
    (tests/golden/make_cli_golden.py; src/main.c:275-503)."""
    cases = _cli_mode_cases()
    assert len(cases) >= 60
    for c in cases:
        pcm, rate, _ = cli_mode_input(c)
        image = ob.encode(pcm, c["bits"], 1024, rate, c["ms"], c["trials"])
        rec, _ = ob.decode(image)
        assert sha256(wav16_bytes(rec, rate)) == c["reconstructed_sha256"], c
        assert sha256(wav16_bytes(ob.residual(pcm, rec), rate)) == c["residual_sha256"], c
        assert ob.stats_line(ob.error_stats(pcm, rec)) == c["stats_line"], c


def test_oracle_on_reference_roundtrip_suite():
    """The reference's own encode->decode suite (test/test_aad_encode_decode.c:283-616: three
    synthetic inputs x 36 parameter sets, six WAV files incl. two real recordings x bits x block
    sizes x M/S) through the oracle: its RMSE bounds hold AND every .aad image / decoded PCM equals
    what the compiled reference produced (tests/golden/roundtrip_suite.json)."""
    from roundtrip_suite import CASES, suite_input, suite_rmse
    assert len(CASES) >= 215
    for c in CASES:
        pcm, file_bytes = suite_input(c)
        image = ob.encode(pcm, c["bits"], c["max_block_size"], c["sampling_rate"], c["ms"], c["trials"])
        dec, _ = ob.decode(image)
        assert sha256(image) == c["aad_sha256"], c
        assert sha256(dec.tobytes()) == c["decoded_sha256"], c
        assert suite_rmse(pcm, dec) < c["rms_epsilon"], c
        if file_bytes is not None:
            assert len(image) < file_bytes // 2, c
