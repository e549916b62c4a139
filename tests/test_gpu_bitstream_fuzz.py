"""Every GPU decode path on bitstreams NO ENCODER WRITES (tests/bitstream_fuzz.py): crafted block headers - step index
on both clamps and up to 4087, shift 0-15 with full-range int16 weights (int32-wrapping predictions), any history -
and crafted code bodies (runs of extreme codes into both clamps of the index walk and both output rails,
alternations, noise), 1-8 channels, 2/3/4 bits, M/S, 1-3 blocks with a ragged last one.

The reference defines a result for any such bytes (src/aad_decoder.c:364-391 header reload, :376 the shift,
:396-451 the code walk).  Pins:
  * tests/golden/bitstream_fuzz.json - 1100 images' decode hashes from the COMPILED reference (800 crafted block contents, 300 file
    headers whose block_size and samples_per_block do not belong together: blocks walked by one, codes read by the other)
    (make_bitstream_golden.py); tests/test_bitstream_fuzz.py holds the oracle to the same hashes on the CPU;
  * the oracle, for the same-format batches built here (checked against the reference in the build container by
    test_bitstream_fuzz.py::test_oracle_matches_compiled_reference_on_fresh_seeds).

Paths: host-memory batches under every lane mapping ("dense", "dense-tiled", "quad" = split decoder with its
residual rows in LDS and through the device scratch buffer, "quad-fused", "auto"), as one tile and cut into
bare-block tiles; device-resident plans with and without the file header (has_file_header = 0: DecodeBlock batched);
the legacy AADDecoder_DecodeWhole / AADDecoder_DecodeBlock.  Bar: bit-exact.
"""
import collections
import functools

import numpy as np
import pytest

import bitstream_fuzz as bf
import oracle_binding as ob
from aad_amd.capi import STREAM_DESC_DTYPE
from aad_amd.engine import parse_header

pytestmark = pytest.mark.gpu

MAPPINGS = ["dense", "dense-tiled", "quad", "quad-fused", "auto"]
GOLDEN = bf.golden_cases()


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def golden_groups():
    """golden cases grouped by format (a host batch is one format), images rebuilt once"""
    groups = collections.OrderedDict()
    for rec in GOLDEN:
        case = bf.case_of_record(rec)
        key = (rec["channels"], rec["bits"], rec["ms"], rec["block_size"], rec["spb"])
        groups.setdefault(key, []).append((rec, case["image"]))
    return groups


def _first_diff(got, want):
    bad = np.argwhere(got != want)
    return {"first bad [sample, channel]": bad[0].tolist(), "count": int(len(bad)),
            "got": got[bad[0][0]:bad[0][0] + 3].tolist(), "want": want[bad[0][0]:bad[0][0] + 3].tolist()}


@pytest.mark.parametrize("tile_kbytes", [0, 1])
@pytest.mark.parametrize("mapping", MAPPINGS)
def test_golden_crafted_images_host_batches(engine, golden_groups, mapping, tile_kbytes):
    """all 800 golden images through AADHip_DecodeBatch == the compiled reference's decode (hashes)"""
    engine.set_mapping(mapping)
    engine.set_tile_kbytes(tile_kbytes)
    try:
        for key, members in golden_groups.items():
            outs = engine.decode_host([img for _, img in members])
            for (rec, img), got in zip(members, outs):
                if bf.pcm_hash(got) != rec["decoded_sha256"]:
                    raise AssertionError((mapping, tile_kbytes, rec["name"], key, rec["header_kind"], rec["body_kind"],
                                          rec["num_samples"], _first_diff(got, bf.oracle_decode(img))))
    finally:
        engine.set_mapping("auto")
        engine.set_tile_kbytes(0)


@functools.lru_cache(maxsize=None)
def _same_format_batch(tag, count, channels, bits, ms, mbs, blocks, last):
    """last: samples in the last block, 0 = a full one"""
    cases = [bf.make_case("%s/%d" % (tag, i), channels=channels, bits=bits, max_block_size=mbs, ms=ms, blocks=blocks,
                          last=last or 1 << 30) for i in range(count)]
    assert len({(c["block_size"], c["spb"], c["num_samples"], len(c["image"])) for c in cases}) == 1
    return cases


# (channels, bits, ms, max_block_size, blocks, last-block samples or 0 = full, streams)
BATCHES = [
    (2, 4, False, 1024, 1, 0, 96),     # BASELINE configs 2/3 geometry, few rows: split decoder keeps its rows in LDS
    (2, 4, False, 1024, 1, 0, 2300),   # 4600 recurrences: split decoder through the scratch buffer; tiled kernel's home ground
    (2, 4, True, 1024, 2, 517, 300),
    (1, 4, False, 1024, 1, 0, 700),
    (1, 4, False, 1024, 3, 9, 130),
    (2, 3, False, 1024, 1, 0, 500),
    (2, 3, True, 512, 2, 100, 260),
    (1, 3, False, 1024, 2, 1000, 333),
    (2, 2, False, 1024, 1, 0, 400),
    (2, 2, True, 777, 3, 3, 150),      # last block: three stored samples only
    (1, 2, False, 1024, 1, 0, 300),    # 4024 coded samples per row: beyond the split decoder's LDS rows
    (1, 4, False, 4096, 2, 5000, 64),  # 8132-sample blocks
    (8, 3, False, 1024, 1, 0, 200),    # BASELINE config 4 geometries
    (8, 2, False, 1024, 2, 77, 120),
    (5, 4, False, 600, 2, 0, 90),
    (3, 2, False, 1024, 1, 0, 64),
]


@pytest.mark.parametrize("mapping", MAPPINGS)
@pytest.mark.parametrize("batch", BATCHES, ids=lambda b: "%dch%db%s_mbs%d_x%d_last%d_n%d" % (b[0], b[1], "ms" if b[2] else "", b[3], b[4], b[5], b[6]))
def test_crafted_same_format_batches_device_plans(engine, mapping, batch):
    """device-resident plans over many crafted streams of one format, with the file header (DecodeWhole batched) and as
    bare blocks (has_file_header = 0, every block its own table entry: DecodeBlock batched) == the oracle"""
    import torch
    channels, bits, ms, mbs, blocks, last, streams = batch
    cases = _same_format_batch("batch-%d-%d-%d-%d-%d-%d" % (channels, bits, ms, mbs, blocks, last), streams, channels, bits, ms, mbs, blocks, last)
    spb, block_size, n, size = cases[0]["spb"], cases[0]["block_size"], cases[0]["num_samples"], len(cases[0]["image"])
    want = np.stack([bf.oracle_decode(c["image"]) for c in cases])                  # [streams, n, ch]
    # PCM runs start on 16-byte boundaries and image pitches are multiples of 128: the layouts on which the tiled kernel applies
    pitch_pcm = -(-(n * channels) // 8) * 8
    pitch_img = -(-size // 128) * 128
    phase = 64 if bits == 3 else 37  # images at an odd byte phase (3-bit rows take the tiled kernel at granule-common phases only)
    flat = np.zeros(phase + streams * pitch_img + 256, dtype=np.uint8)
    for i, c in enumerate(cases):
        flat[phase + i * pitch_img:phase + i * pitch_img + size] = np.frombuffer(c["image"], dtype=np.uint8)
    d_img = torch.from_numpy(flat).cuda()
    hd = parse_header(cases[0]["image"][:31])

    whole = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
    whole["pcm_offset"] = np.arange(streams, dtype=np.uint64) * np.uint64(pitch_pcm)
    whole["data_offset"] = phase + np.arange(streams, dtype=np.uint64) * np.uint64(pitch_img)
    whole["data_size"], whole["num_samples"] = size, n

    # every block as its own stream of a headerless plan
    bare = np.zeros(streams * blocks, dtype=STREAM_DESC_DTYPE)
    for k in range(blocks):
        nk = min(spb, n - k * spb)
        sel = slice(k, None, blocks)
        bare["pcm_offset"][sel] = whole["pcm_offset"] + np.uint64(k * spb * channels)
        bare["data_offset"][sel] = whole["data_offset"] + np.uint64(31 + k * block_size)
        bare["data_size"][sel] = min(block_size, size - 31 - k * block_size)
        bare["num_samples"][sel] = nk

    engine.set_mapping(mapping)
    try:
        for label, table, with_header in (("whole", whole, True), ("bare blocks", bare, False)):
            plan = engine.decode_plan(hd, table, with_header)
            out = torch.full((streams * pitch_pcm + 64,), 0x5A5A, dtype=torch.int16, device="cuda")
            plan.run(d_img, out)
            torch.cuda.synchronize()
            plan.close()
            host = out.cpu().numpy()
            got = host[:streams * pitch_pcm].reshape(streams, pitch_pcm)[:, :n * channels].reshape(streams, n, channels)
            if not np.array_equal(got, want):
                s = int(np.argwhere((got != want).any(axis=(1, 2)))[0][0])
                raise AssertionError((mapping, label, batch, "stream", s, cases[s]["header_kind"], cases[s]["body_kind"],
                                      _first_diff(got[s], want[s])))
            # nothing outside the streams' runs is written
            pad = host[:streams * pitch_pcm].reshape(streams, pitch_pcm)[:, n * channels:]
            assert (pad == 0x5A5A).all() and (host[streams * pitch_pcm:] == 0x5A5A).all(), (mapping, label, batch, "wrote outside its rows")
    finally:
        engine.set_mapping("auto")


def test_legacy_api_on_crafted_images():
    """AADDecoder_DecodeWhole and AADDecoder_DecodeBlock of libaad_hip.so (the reference's own entry points,
    src/aad_decoder.c:478-538, :321-475) on the golden images of up to two channels == the compiled reference's hashes"""
    import aad_amd
    codec = aad_amd.LegacyCodec(aad_amd.load_library())
    done = 0
    for rec in GOLDEN[:600:3]:
        case = bf.case_of_record(rec)
        img = case["image"]
        pcm, hd = codec.decode(img)
        assert bf.pcm_hash(pcm) == rec["decoded_sha256"], ("DecodeWhole", rec["name"])
        if done % 4 == 0:  # block by block through AADDecoder_DecodeBlock
            pos, left, parts = 31, rec["num_samples"], []
            while left > 0:
                nk = min(left, rec["spb"])
                block = img[pos:pos + rec["block_size"]]
                parts.append(codec.decode_block(hd, block, nk))
                pos += len(block)
                left -= nk
            assert bf.pcm_hash(np.concatenate(parts)) == rec["decoded_sha256"], ("DecodeBlock", rec["name"])
        done += 1
    assert done == 200
    geometry = [rec for rec in GOLDEN if rec["header_kind"] == "geometry"][::4]
    for rec in geometry:  # inconsistent block_size / samples_per_block: AADDecoder_DecodeWhole walks by one and reads by the other
        pcm, _ = codec.decode(bf.case_of_record(rec)["image"])
        assert bf.pcm_hash(pcm) == rec["decoded_sha256"], ("DecodeWhole, geometry", rec["name"], rec["block_size"], rec["spb"], rec["fits"])
    assert len(geometry) == 75


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_inconsistent_geometry_up_to_eight_channels(engine, mapping):
    """block_size and samples_per_block that do not belong together, 1-8 channels (the reference-pinned 1-2 channel cases are in the
    golden set above; wider ones against the oracle, which the build container holds to the reference on this input class)"""
    engine.set_mapping(mapping)
    try:
        for tile in (0, 1):
            engine.set_tile_kbytes(tile)
            for name in bf.case_names(60, "gw"):
                case = bf.make_geometry_case(name, max_channels=8)
                got = engine.decode_host([case["image"]])[0]
                want = bf.oracle_decode(case["image"])
                assert np.array_equal(got, want), (mapping, tile, name, case["channels"], case["bits"], case["block_size"], case["spb"], case["fits"],
                                                   case["blocks"], _first_diff(got, want))
    finally:
        engine.set_mapping("auto")
        engine.set_tile_kbytes(0)


def test_header_index_field_beyond_the_table(engine):
    """index fields 4088..4095 (the reference reads past its table: undefined there) decode as 4087 under every mapping,
    and nothing faults"""
    case = bf.make_case("index-field", channels=2, bits=4, max_block_size=1024, ms=False, blocks=2)
    img = bytearray(case["image"])
    images = []
    for idx in (4087, 4088, 4095):
        for c in range(2):
            for k in range(2):
                o = 31 + k * case["block_size"] + 18 * c
                img[o], img[o + 1] = idx >> 4, ((idx & 15) << 4) | (img[o + 1] & 15)
        images.append(bytes(img))
    want = bf.oracle_decode(images[0])
    for mapping in MAPPINGS:
        engine.set_mapping(mapping)
        try:
            for got in engine.decode_host(images):
                assert np.array_equal(got, want), mapping
        finally:
            engine.set_mapping("auto")
