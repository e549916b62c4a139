"""The dense encoders' byte ring (aad_amd/csrc/aad_encode.hip.h ByteRing): device-resident plans whose images start on
64-byte boundaries - the layouts on which the dense 4- and 2-bit encoders append their output (file header, block
headers, packed codes, tail units) to a per-row ring in LDS and store every 64-byte sector of an image once, whole.
Against the oracle, byte for byte, INCLUDING the bytes around every image (the buffer is pre-filled with a pattern: a
sector store must not reach past an image's data_size, and nothing may land in front of an image).
Covered: mono / stereo x 4- / 2-bit x M/S, one sample to several blocks per stream, ragged last blocks and tail units,
uniform tables and shuffled ones, data_size == the encoded size (the last sector goes out byte by byte) and padded
to the next sector (whole-sector stores), odd block sizes, more rows than a wave and fewer."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import STREAM_DESC_DTYPE, make_parameter
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("uniform", [True, False])
def test_ring_encoder_matches_oracle(engine, bits, channels, uniform, monkeypatch):
    import torch
    monkeypatch.setenv("AAD_HIP_ENCODE_RING", "2")  # every geometry that can (the host's policy leaves stereo 2-bit on its per-lane stores)
    rng = np.random.default_rng(8800 + 10 * bits + channels + (100 if uniform else 0) + 1000 * int(os.environ.get("AAD_TEST_SEED_OFFSET", "0")))
    engine.set_mapping("dense")
    try:
        for rep in range(10):
            ms = bool(channels == 2 and rep % 3 == 1)
            mbs = int(rng.choice([1024, 1024, 1024, 512, 300, 2048, 777]))
            rc, block_size, spb = ob.geometry(mbs, channels, bits)
            assert rc == 0
            streams = int(rng.choice([1, 3, 31, 32, 33, 64, 65, 130]))
            if uniform:
                lengths = [int(rng.choice([spb, 2 * spb, 3 * spb + 17, spb - 5, 2 * spb + 13, 9, 21, 4, 1, 5]))] * streams
            else:
                lengths = [int(rng.choice([spb, 2 * spb, int(rng.integers(1, 3 * spb + 20)), int(rng.integers(1, 40))])) for _ in range(streams)]
            pcms = [synth_pcm(1, n, channels, seed=int(rng.integers(0, 1 << 30)), kind=str(rng.choice(["music", "noise"])))[0] for n in lengths]
            param = make_parameter(channels, bits, mbs, 48000, ms, 0)
            want = [ob.encode(p, bits, mbs, 48000, ms, 0) for p in pcms]
            sizes = [len(w) for w in want]
            padded = bool(rep % 2)  # data_size: the encoded size itself, or rounded up to the sector
            d = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
            order = list(range(streams)) if uniform else list(rng.permutation(streams))
            pitch = -(-(max(sizes) + int(rng.integers(0, 200))) // 64) * 64
            pos_p, pos_d = int(rng.integers(0, 16)), 64 * int(rng.integers(0, 3))
            for slot in order:
                d["pcm_offset"][slot], d["data_offset"][slot] = pos_p, pos_d
                d["data_size"][slot] = -(-sizes[slot] // 64) * 64 if padded else sizes[slot]
                d["num_samples"][slot] = lengths[slot]
                pos_p += (max(lengths) if uniform else lengths[slot]) * channels + int(rng.integers(0, 7))
                pos_d += pitch if uniform else -(-(sizes[slot] + int(rng.integers(0, 130))) // 64) * 64
            flat = np.zeros(pos_p + 64, dtype=np.int16)
            for i, p in enumerate(pcms):
                o = int(d["pcm_offset"][i])
                flat[o:o + p.size] = p.reshape(-1)
            d_pcm = torch.from_numpy(flat).cuda()
            d_img = torch.full((pos_d + 256,), 0xA5, dtype=torch.uint8, device="cuda")
            plan = engine.encode_plan(param, d)
            plan.run(d_pcm, d_img, None)
            torch.cuda.synchronize()
            plan.close()
            img = d_img.cpu().numpy()
            expect = np.full(pos_d + 256, 0xA5, dtype=np.uint8)
            for i, w in enumerate(want):
                o = int(d["data_offset"][i])
                expect[o:o + len(w)] = np.frombuffer(w, dtype=np.uint8)
            # inside an image's data_size but behind its last byte the encoder may leave anything it likes; everything
            # else - the images and every byte outside them - is pinned
            free = np.zeros(pos_d + 256, dtype=bool)
            for i in range(streams):
                o = int(d["data_offset"][i])
                free[o + sizes[i]:o + int(d["data_size"][i])] = True
            bad = np.nonzero((img != expect) & ~free)[0]
            assert bad.size == 0, (bits, channels, uniform, rep, mbs, ms, streams, lengths[:4], padded, "first bad byte at", int(bad[0]))
    finally:
        engine.set_mapping("auto")


@pytest.mark.parametrize("bits,channels", [(4, 2), (4, 1), (3, 2), (3, 1), (2, 2), (2, 1)])
def test_ring_encoder_chip_filling_batch(engine, bits, channels, monkeypatch):
    """a batch big enough for "auto" to take the dense encoders (workgroups of four waves: four waves of rows share the ring
    area), two blocks per stream - against the oracle on a sample of streams, and every repetition of the tile against the first"""
    import torch
    monkeypatch.setenv("AAD_HIP_ENCODE_RING", "2")
    spb = {4: 1984, 3: 2632, 2: 3960}[bits] // channels
    streams = 70000 // channels
    param = make_parameter(channels, bits, 1024, 48000, False, 0)
    tile = torch.from_numpy(synth_pcm(300, 2 * spb, channels, seed=77)).cuda()
    pcm = tile.repeat((-(-streams // 300), 1, 1))[:streams].contiguous()
    img, size = engine.encode_uniform(pcm, param)
    torch.cuda.synchronize()
    host = img[:300].cpu().numpy()
    src = tile.cpu().numpy()
    for s in range(0, 300, 37):
        assert bytes(host[s, :size]) == ob.encode(src[s], bits, 1024, 48000, False, 0), s
    assert torch.equal(img[:300, :size], img[300:600, :size])
