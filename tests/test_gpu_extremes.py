"""Extremes of the parameter space on the GPU, against the oracle (which tests/test_oracle_extremes.py holds to the compiled
reference in the build container):
  * the largest block sizes the format allows (max_block_size is a uint16_t: 65 535 -> up to 262 072 samples per block,
    reference src/aad_encoder.c:85-131), encode and decode under every mapping, 1 / 2 / 8 channels, with and without the trial
    search - per-block counters, LDS rows (the split decoder's fit 2048 samples) and tile cuts all see their largest values;
  * AADDecoder_DecodeBlock with a buffer shorter or longer than the block ("decode until the buffer is full",
    src/aad_decoder.c:356-358, :386-391 for fewer than four samples)."""
import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import make_parameter
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("mapping", ["auto", "dense", "dense-tiled", "quad", "quad-fused"])
@pytest.mark.parametrize("mbs", [65535, 65534, 40001])
def test_largest_block_sizes(engine, mapping, mbs):
    engine.set_mapping(mapping)
    try:
        for ch, bits, trials, ms in ((1, 4, 0, False), (1, 2, 1, False), (2, 3, 0, True), (2, 4, 2, False), (8, 2, 0, False), (1, 3, 0, False)):
            rc, block_size, spb = ob.geometry(mbs, ch, bits)
            assert rc == 0
            lengths = [int(spb * 1.3) + 7, spb, 5, spb + 1]
            pcms = [synth_pcm(1, n, ch, seed=mbs % 1000 + 10 * ch + bits + i, kind=("music", "noise")[i % 2])[0] for i, n in enumerate(lengths)]
            param = make_parameter(ch, bits, mbs, 48000, ms, trials)
            want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
            for tile in (0, 64):
                engine.set_tile_kbytes(tile)
                got = engine.encode_host(pcms, param)
                for i, (g, w) in enumerate(zip(got, want)):
                    assert g == w, (mapping, mbs, ch, bits, trials, ms, tile, lengths[i])
                dec = engine.decode_host(want)
                for i, (d, w) in enumerate(zip(dec, want)):
                    assert np.array_equal(d, ob.decode(w)[0]), (mapping, mbs, ch, bits, tile, lengths[i])
    finally:
        engine.set_mapping("auto")
        engine.set_tile_kbytes(0)


def test_decode_block_buffer_shorter_and_longer_than_the_block():
    import aad_amd
    codec = aad_amd.LegacyCodec(aad_amd.load_library())
    rng = np.random.default_rng(77)
    for ch, bits, ms, mbs in ((2, 4, False, 1024), (1, 3, False, 300), (2, 2, True, 512), (1, 4, False, 4096)):
        _, block_size, spb = ob.geometry(mbs, ch, bits)
        pcm = synth_pcm(1, 2 * spb + 9, ch, seed=int(rng.integers(0, 1 << 20)))[0]
        img = ob.encode(pcm, bits, mbs, 48000, ms, 0)
        hd = codec.decode_header(img)
        full = ob.decode(img)[0]
        for b in range(3):
            blk = img[31 + b * block_size: 31 + (b + 1) * block_size]
            have = min(spb, len(full) - b * spb)  # samples this block carries
            for want_samples in sorted({1, 2, 3, 4, 5, 8, 9, int(rng.integers(6, spb)), spb, spb + 100}):
                if want_samples > have and b == 2:
                    continue  # the short last block holds no codes beyond its own samples (the reference would read past it)
                got = codec.decode_block(hd, blk, want_samples)
                n = min(want_samples, spb)
                assert got.shape[0] == n, (ch, bits, b, want_samples, got.shape)
                if ms and n < 4:
                    # fewer than four frames: the reference emits the header's stored samples (:386-391) but applies the M/S
                    # inverse to `n` frames only; the oracle restates that
                    pass
                assert np.array_equal(got, _oracle_block(hd, blk, want_samples, ch)), (ch, bits, ms, b, want_samples)


def _oracle_block(hd, blk, want_samples, ch):
    import ctypes as C
    ohd = ob.AadoHeader(hd.format_version, hd.codec_version, hd.num_channels, hd.num_samples, hd.sampling_rate, hd.bits_per_sample,
                        hd.block_size, hd.num_samples_per_block, hd.ch_process_method)
    out = np.zeros((max(want_samples, 4), ch), dtype=np.int16)
    frames = C.c_uint32(0)
    bb = np.frombuffer(blk, dtype=np.uint8)
    assert ob.lib().aado_decode_block(C.byref(ohd), bb.ctypes.data, len(bb), out.ctypes.data, want_samples, C.byref(frames)) == 0
    return out[:frames.value]
