"""The sector-tiled dense kernels (aad_amd/csrc/aad_decode_tiled.hip.h): device-resident plans whose PCM is
16-byte aligned - the layouts on which the host launches them when the mapping is "dense-tiled" (and, for
chip-filling batches, "auto") - against the oracle and against the per-lane dense kernels, bit for bit.
Covered: 4-, 3- and 2-bit codes (3-bit: image pitches a multiple of the granule, any common phase) x mono / stereo x M/S, uniform tables (the table-free path) and shuffled ones,
images at every byte phase (0..127: the code bytes' phase inside a sector is what the tiles absorb), one to
several blocks per stream with ragged last blocks, streams shorter than a chunk, odd block sizes, images cut
short (missing bytes decode as zero), more rows than one wave and fewer than one."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import STREAM_DESC_DTYPE, make_parameter
from aad_amd.engine import parse_header
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _decode_with(engine, mapping, header, desc, d_img, n_pcm):
    import torch
    engine.set_mapping(mapping)
    try:
        plan = engine.decode_plan(header, desc, True)
        out = torch.full((n_pcm,), 0x5A5A, dtype=torch.int16, device="cuda")  # a pattern: bytes the kernel must not touch stay
        plan.run(d_img, out)
        torch.cuda.synchronize()
        plan.close()
        return out.cpu().numpy()
    finally:
        engine.set_mapping("auto")


def _layout(rng, lengths, sizes, ch, uniform, pad_dat, gran=1):
    """stream table with every PCM start on a 16-byte boundary (8 int16) and images at byte phase pad_dat; gran: the images'
    pitch is a multiple of it (3-bit rows take the tiled kernel only when all code bytes share their phase inside a granule)"""
    streams = len(lengths)
    d = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
    order = list(range(streams)) if uniform else list(rng.permutation(streams))
    pitch_pcm = -(-(max(lengths) * ch + int(rng.integers(0, 40))) // 8) * 8
    pitch_dat = -(-(max(sizes) + int(rng.integers(0, 70))) // gran) * gran
    pos_p, pos_d = 8 * int(rng.integers(0, 9)), pad_dat
    for slot in order:
        d["pcm_offset"][slot], d["data_offset"][slot] = pos_p, pos_d
        d["data_size"][slot], d["num_samples"][slot] = sizes[slot], lengths[slot]
        pos_p += pitch_pcm if uniform else -(-(lengths[slot] * ch + int(rng.integers(0, 30))) // 8) * 8
        pos_d += pitch_dat if uniform else -(-(sizes[slot] + int(rng.integers(0, 90))) // gran) * gran
    return d, pos_p, pos_d


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("uniform", [True, False])
def test_tiled_decoder_matches_oracle_and_dense(engine, bits, channels, uniform):
    import torch
    rng = np.random.default_rng(7700 + 10 * bits + channels + (100 if uniform else 0) + 1000 * int(os.environ.get("AAD_TEST_SEED_OFFSET", "0")))
    for rep in range(10):
        ms = bool(channels == 2 and rep % 3 == 1)
        mbs = int(rng.choice([1024, 1024, 1024, 512, 300, 2048, 777]))
        if bits == 3 and rep % 2 == 0:
            mbs = int(rng.choice([1024, 512, 2048]))  # multi-block 3-bit streams stay on the tiled kernel when the block size is a granule multiple
        rc, block_size, spb = ob.geometry(mbs, channels, bits)
        assert rc == 0
        streams = int(rng.choice([1, 3, 31, 32, 33, 64, 65, 130]))
        if uniform:
            n0 = int(rng.choice([spb, 2 * spb, 3 * spb + 17, spb - 5, 2 * spb + 13, 9, 21, 4, 1]))
            lengths = [n0] * streams
        else:
            lengths = [int(rng.choice([spb, 2 * spb, int(rng.integers(1, 3 * spb + 20)), int(rng.integers(1, 40))])) for _ in range(streams)]
        pcms = [synth_pcm(1, n, channels, seed=int(rng.integers(0, 1 << 30)), kind=str(rng.choice(["music", "noise"])))[0] for n in lengths]
        images = [ob.encode(p, bits, mbs, 48000, ms, 0) for p in pcms]
        sizes = [len(w) for w in images]
        if rep % 4 == 3:  # some images cut short inside their last block: the missing bytes decode as zero
            for i in range(0, streams, 3):
                blocks = -(-lengths[i] // spb)
                last = 31 + (blocks - 1) * block_size
                keep = last + 18 * channels + int(rng.integers(0, max(1, sizes[i] - last - 18 * channels)))
                sizes[i] = min(sizes[i], keep)
        d, n_pcm, n_dat = _layout(rng, lengths, sizes, channels, uniform, int(rng.integers(0, 128)), 64 * channels if bits == 3 else 1)
        flat = np.zeros(n_dat + 256, dtype=np.uint8)
        for i, w in enumerate(images):
            o = int(d["data_offset"][i])
            flat[o:o + sizes[i]] = np.frombuffer(w, dtype=np.uint8)[:sizes[i]]
        d_img = torch.from_numpy(flat).cuda()
        hd = parse_header(images[0][:31])
        tiled = _decode_with(engine, "dense-tiled", hd, d, d_img, n_pcm + 64)
        dense = _decode_with(engine, "dense", hd, d, d_img, n_pcm + 64)
        if not np.array_equal(tiled, dense):
            bad = np.nonzero(tiled != dense)[0]
            first = int(bad[0])
            owner = max((i for i in range(streams) if int(d["pcm_offset"][i]) <= first), key=lambda i: int(d["pcm_offset"][i]), default=-1)
            rel = first - int(d["pcm_offset"][owner]) if owner >= 0 else first
            raise AssertionError((bits, channels, uniform, rep, mbs, streams, lengths[:4], "tiled != dense (untouched bytes included)",
                                  "first bad int16", first, "count", int(bad.size), "stream", owner, "offset in stream", rel,
                                  "block", rel // (spb * channels), "in block", rel % (spb * channels), "length", lengths[owner], "size", sizes[owner],
                                  "tiled", tiled[first:first + 6].tolist(), "dense", dense[first:first + 6].tolist()))
        for i, w in enumerate(images):
            want = np.zeros((lengths[i], channels), dtype=np.int16)
            buf = np.frombuffer(w, dtype=np.uint8)[:sizes[i]].copy()
            ob.lib().aado_decode_stream(buf.ctypes.data, len(buf), 8, want.ctypes.data, lengths[i], None)
            o = int(d["pcm_offset"][i])
            got = tiled[o:o + lengths[i] * channels].reshape(-1, channels)
            if sizes[i] == len(w):
                assert np.array_equal(got, want), (bits, channels, uniform, rep, mbs, i, lengths[i])
            else:  # a cut image: whole blocks in front of the cut are exact (what the engine documents for truncated images)
                blocks_ok = (sizes[i] - 31) // block_size
                assert np.array_equal(got[:blocks_ok * spb], want[:blocks_ok * spb]), (bits, channels, uniform, rep, mbs, i, "cut")


@pytest.mark.parametrize("bits,channels", [(4, 2), (4, 1), (3, 2), (3, 1), (2, 2), (2, 1)])
def test_tiled_decoder_chip_filling_uniform_batch(engine, bits, channels):
    """the shape bench.py's `saturated` leg runs (one-block streams, images pitched at multiples of 64 bytes), big enough
    for "auto" to take the tiled kernel: auto == dense-tiled == dense, and the decode reproduces the PCM's encode"""
    import torch
    samples = {4: 1984, 3: 2632, 2: 3960}[bits] // channels
    streams = (400000 if (bits, channels) == (4, 2) else 70000) // channels  # beyond the auto threshold of the 4-bit geometries
    param = make_parameter(channels, bits, 1024, 48000, False, 0)
    tile = torch.from_numpy(synth_pcm(500, samples, channels, seed=4321)).cuda()
    pcm = tile.repeat((-(-streams // 500), 1, 1))[:streams].contiguous()
    img, size = engine.encode_uniform(pcm, param)
    outs = {}
    for mapping in ("auto", "dense-tiled", "dense"):
        engine.set_mapping(mapping)
        try:
            outs[mapping], _ = engine.decode_uniform(img, size)
            torch.cuda.synchronize()
        finally:
            engine.set_mapping("auto")
    assert torch.equal(outs["auto"], outs["dense"]) and torch.equal(outs["dense-tiled"], outs["dense"])
    host = img[:3].cpu().numpy()
    for s in range(3):
        assert np.array_equal(outs["dense-tiled"][s].cpu().numpy(), ob.decode(bytes(host[s, :size]))[0])
