"""The ORACLE on bitstreams no encoder writes (tests/bitstream_fuzz.py): random / adversarial block
headers (step index on both clamps and up to 4087, shift 0-15 with full-range int16 weights, any
history) and code bodies (runs of extreme codes, alternations, noise), 1-8 channels, 2/3/4 bits,
M/S, 1-3 blocks with a ragged last one.

  * against tests/golden/bitstream_fuzz.json - decode hashes of the COMPILED reference
    (AADDecoder_DecodeWhole, src/aad_decoder.c:478-538; generator make_bitstream_golden.py).  This
    pin travels to the GPU box, where tests/test_gpu_bitstream_fuzz.py runs the same images through
    every GPU decode path;
  * `ref`-marked: fresh seeds against the compiled reference itself (build container only).

Round 4, second family: file headers whose block_size and samples_per_block do not belong together (the reference's header
checks relate neither to the other, src/aad_decoder.c:173-225): blocks are walked by block_size, codes read by
samples_per_block - on into the following blocks' bytes when a block is too small for its samples (found there: the oracle
read zeros past the block's end; the reference reads the file's next bytes).

Found by this file's first run: the oracle (and the three GPU decoders) clamped a block header's
step index to 4080; the reference takes the field as it is (:365-366), so 4081..4087 - same table
slot, 255 - continue the index walk from the unclamped value.
"""
import numpy as np
import pytest

import bitstream_fuzz as bf
import oracle_binding as ob

GOLDEN = bf.golden_cases()


def test_golden_covers_the_input_classes():
    keys = {(r["header_kind"], r["body_kind"]) for r in GOLDEN if r["header_kind"] != "geometry"}
    assert len(keys) == len(bf.HEADER_KINDS) * len(bf.BODY_KINDS)
    geo = [r for r in GOLDEN if r["header_kind"] == "geometry"]
    assert len(geo) == 300 and sum(r["spb"] > r["fits"] for r in geo) > 60 and sum(r["spb"] < 4 for r in geo) > 10
    assert {r["bits"] for r in GOLDEN} == {2, 3, 4}
    assert {r["channels"] for r in GOLDEN} == set(range(1, 9))
    assert any(r["ms"] for r in GOLDEN)
    assert len(GOLDEN) == 1100


@pytest.mark.parametrize("part", range(8))
def test_oracle_matches_reference_hashes(part):
    for rec in GOLDEN[part::8]:
        case = bf.case_of_record(rec)
        assert bf.pcm_hash(bf.oracle_decode(case["image"])) == rec["decoded_sha256"], rec["name"]


def test_crafted_streams_reach_the_corners():
    """the generator does what its docstring says: both rails, both index clamps, wrapping predictions"""
    lo = hi = rails = wraps = top = 0
    for rec in GOLDEN[:300]:
        assert rec["header_kind"] != "geometry"
        case = bf.case_of_record(rec)
        img, ch = case["image"], case["channels"]
        pcm = bf.oracle_decode(img)
        rails += int((pcm == 32767).any() and (pcm == -32768).any())
        for c in range(ch):
            v = int.from_bytes(img[31 + 18 * c:33 + 18 * c], "big")
            lo += (v >> 4) == 0
            hi += (v >> 4) == 4080
            top += (v >> 4) > 4080
            w = [int.from_bytes(img[33 + 18 * c + 4 * k:35 + 18 * c + 4 * k], "big", signed=True) << (v & 15) for k in range(4)]
            h = [int.from_bytes(img[35 + 18 * c + 4 * k:37 + 18 * c + 4 * k], "big", signed=True) for k in range(4)]
            wraps += abs(16384 + sum(a * b for a, b in zip(w, h))) >= 1 << 31
    assert lo > 10 and hi > 10 and top > 10 and rails > 50 and wraps > 100, (lo, hi, top, rails, wraps)


def test_header_index_above_table_is_defined_here():
    """4088..4095 in the header's index field: the reference reads past its step table (undefined); here it is 4087"""
    case = bf.make_case("index-field", channels=1, bits=4, max_block_size=256)
    img = bytearray(case["image"])
    shift = img[32] & 15
    outs = []
    for idx in (4087, 4088, 4095):
        img[31], img[32] = idx >> 4, ((idx & 15) << 4) | shift
        outs.append(bf.oracle_decode(bytes(img)))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.ref
@pytest.mark.parametrize("part", range(4))
def test_oracle_matches_compiled_reference_on_fresh_seeds(part):
    import aad_amd
    ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
    for name in bf.case_names(400, "r%d-" % part):
        case = bf.make_case(name)
        want, _ = ref.decode(case["image"])
        assert np.array_equal(bf.oracle_decode(case["image"]), want), name
    for name in bf.case_names(400, "g%d-" % part):  # inconsistent block_size / samples_per_block
        case = bf.make_geometry_case(name)
        want, _ = ref.decode(case["image"])
        assert np.array_equal(bf.oracle_decode(case["image"]), want), name
    for i, name in enumerate(bf.case_names(60, "w%d-" % part)):
        case = bf.make_case(name, channels=3 + i % 6)
        want = np.concatenate([ref.decode(bf.channel_as_mono_image(case, c))[0] for c in range(case["channels"])], axis=1)
        assert np.array_equal(bf.oracle_decode(case["image"]), want), name


@pytest.mark.ref
def test_reference_decode_block_on_crafted_blocks():
    """AADDecoder_DecodeBlock directly (src/aad_decoder.c:321-475) on the crafted blocks, next to the oracle's block decode"""
    import ctypes as C
    import aad_amd
    ref = aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))
    for name in bf.case_names(120, "b"):
        case = bf.make_case(name)
        hd = ref.decode_header(case["image"])
        pos, left = 31, case["num_samples"]
        ohd = ob.AadoHeader()
        buf = np.frombuffer(case["image"], dtype=np.uint8)
        assert ob.lib().aado_get_header(buf.ctypes.data, len(buf), C.byref(ohd)) == 0
        while left > 0:
            n = min(left, case["spb"])
            block = case["image"][pos:pos + case["block_size"]]
            want = ref.decode_block(hd, block, n)
            got = np.zeros((n, case["channels"]), dtype=np.int16)
            frames = C.c_uint32(0)
            bb = np.frombuffer(block, dtype=np.uint8)
            assert ob.lib().aado_decode_block(C.byref(ohd), bb.ctypes.data, len(bb), got.ctypes.data, n, C.byref(frames)) == 0
            assert frames.value == n and np.array_equal(got, want), (name, pos)
            pos += len(block)
            left -= n
