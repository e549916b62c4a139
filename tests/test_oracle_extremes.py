"""The oracle against the compiled reference at the extremes tests/test_gpu_extremes.py runs on the GPU (build container only):
the largest block sizes (max_block_size 65 535 / 65 534 / 40 001) and AADDecoder_DecodeBlock with buffers shorter and longer
than the block (reference src/aad_decoder.c:356-358, :386-391)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def ref():
    import aad_amd
    return aad_amd.LegacyCodec(aad_amd.load_library(ob.REF_SO, hip=False))


@pytest.mark.parametrize("mbs", [65535, 65534, 40001])
def test_largest_block_sizes(ref, mbs):
    for ch in (1, 2):
        for bits in (4, 3, 2):
            _, _, spb = ob.geometry(mbs, ch, bits)
            for n, trials, ms in ((int(spb * 1.3) + 7, 0, False), (spb + 1, 1, ch == 2), (5, 2, False)):
                pcm = synth_pcm(1, n, ch, seed=mbs % 1000 + 10 * ch + bits, kind="music")[0]
                a = ref.encode(pcm, bits, mbs, 48000, ms, trials)
                assert ob.encode(pcm, bits, mbs, 48000, ms, trials) == a, (mbs, ch, bits, n, trials)
                assert np.array_equal(ob.decode(a)[0], ref.decode(a)[0])


def test_decode_block_buffer_shorter_and_longer_than_the_block(ref):
    rng = np.random.default_rng(77)
    for ch, bits, ms, mbs in ((2, 4, False, 1024), (1, 3, False, 300), (2, 2, True, 512), (1, 4, False, 4096)):
        _, block_size, spb = ob.geometry(mbs, ch, bits)
        pcm = synth_pcm(1, 2 * spb + 9, ch, seed=int(rng.integers(0, 1 << 20)))[0]
        img = ref.encode(pcm, bits, mbs, 48000, ms, 0)
        hd = ref.decode_header(img)
        ohd = ob.AadoHeader(hd.format_version, hd.codec_version, hd.num_channels, hd.num_samples, hd.sampling_rate, hd.bits_per_sample,
                            hd.block_size, hd.num_samples_per_block, hd.ch_process_method)
        for b in range(2):  # full blocks: every read stays inside the block
            blk = img[31 + b * block_size: 31 + (b + 1) * block_size]
            for want_samples in sorted({1, 2, 3, 4, 5, 8, 9, int(rng.integers(6, spb)), spb, spb + 100}):
                want = ref.decode_block(hd, blk, want_samples)
                out = np.zeros((max(want_samples, 4), ch), dtype=np.int16)
                frames = C.c_uint32(0)
                bb = np.frombuffer(blk, dtype=np.uint8)
                assert ob.lib().aado_decode_block(C.byref(ohd), bb.ctypes.data, len(bb), out.ctypes.data, want_samples, C.byref(frames)) == 0
                assert frames.value == want.shape[0] == min(want_samples, spb)
                assert np.array_equal(out[:frames.value], want), (ch, bits, ms, b, want_samples)
