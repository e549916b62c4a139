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


SEEDS = int(os.environ.get("AAD_FUZZ_SEEDS", "8"))  # a one-off deeper run: AAD_FUZZ_SEEDS=200 pytest -m gpu tests/test_gpu_fuzz.py


@pytest.mark.parametrize("mapping", ["dense", "dense-tiled", "quad", "quad-fused", "auto"])
@pytest.mark.parametrize("seed", range(SEEDS))
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
            # the host-memory path as one tile, or cut into groups of streams and tiles of blocks
            engine.set_tile_kbytes(int(rng.choice([0, 0, 1, 5, 40])))
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
        engine.set_tile_kbytes(0)


@pytest.mark.parametrize("mapping", ["dense", "dense-tiled", "quad", "quad-fused"])
@pytest.mark.parametrize("seed", range(max(4, SEEDS // 2)))
def test_random_device_resident_plans(engine, mapping, seed):
    """Device-resident plans with hand-made stream tables: uniform layouts at odd alignments (the
    table-free kernel path, the dense decoder's 16-frame lead chunk, streamed stores on and off) and
    shuffled non-uniform ones, multi-block streams included - against the oracle."""
    import torch
    from aad_amd.capi import STREAM_DESC_DTYPE
    from aad_amd.engine import parse_header
    rng = np.random.default_rng(5000 + seed)
    engine.set_mapping(mapping)
    try:
        for _ in range(3):
            ch = int(rng.choice([1, 2, 2, 2]))
            bits = int(rng.choice([4, 4, 3, 2]))
            ms = bool(ch == 2 and rng.integers(0, 4) == 0)
            trials = int(rng.choice([0, 0, 0, 1, 2]))
            mbs = int(rng.choice([1024, 1024, 256, 700]))
            rc, block_size, spb = ob.geometry(mbs, ch, bits)
            assert rc == 0
            streams = int(rng.integers(1, 40))
            uniform = bool(rng.integers(0, 2))
            n0 = int(rng.integers(1, 3 * spb + 20))
            lengths = [n0] * streams if uniform else [int(rng.integers(1, 3 * spb + 20)) for _ in range(streams)]
            pcms = [synth_pcm(1, n, ch, seed=int(rng.integers(0, 1 << 30)), kind=str(rng.choice(["music", "noise"])))[0] for n in lengths]
            param = make_parameter(ch, bits, mbs, 48000, ms, trials)
            sizes = [engine.encoded_size(param, n) for n in lengths]
            # layout: a leading pad of 0..63 int16 / bytes, then streams back to back at a pitch of
            # the (padded) maximum for uniform batches, in shuffled order otherwise
            pad_pcm, pad_dat = int(rng.integers(0, 64)), int(rng.integers(0, 64))
            pitch_pcm = (max(lengths) * ch + int(rng.integers(0, 9))) if uniform else None
            pitch_dat = (max(sizes) + int(rng.integers(0, 17))) if uniform else None
            order = list(range(streams)) if uniform else list(rng.permutation(streams))
            granule = bool(rng.integers(0, 2))
            if granule:
                # images on 64-byte boundaries: the dense stereo 4-bit encoder then stores whole granules,
                # holding three bytes back per group (aad_encode.hip.h run_block) - every other draw
                pad_dat = 0
                if uniform:
                    pitch_dat = -(-pitch_dat // 64) * 64
            d = np.zeros(streams, dtype=STREAM_DESC_DTYPE)
            pos_p, pos_d = pad_pcm, pad_dat
            for slot in order:
                d["pcm_offset"][slot], d["data_offset"][slot] = pos_p, pos_d
                d["data_size"][slot], d["num_samples"][slot] = sizes[slot], lengths[slot]
                pos_p += pitch_pcm if uniform else lengths[slot] * ch + int(rng.integers(0, 5))
                pos_d += pitch_dat if uniform else sizes[slot] + int(rng.integers(0, 7))
                if granule and not uniform:
                    pos_d = -(-pos_d // 64) * 64
            flat = np.zeros(pos_p + 64, dtype=np.int16)
            for i, p in enumerate(pcms):
                o = int(d["pcm_offset"][i])
                flat[o:o + p.size] = p.reshape(-1)
            d_pcm = torch.from_numpy(flat).cuda()
            d_img = torch.zeros(pos_d + 64, dtype=torch.uint8, device="cuda")
            plan = engine.encode_plan(param, d)
            plan.run(d_pcm, d_img, None)
            torch.cuda.synchronize()
            plan.close()
            img = d_img.cpu().numpy()
            want = [ob.encode(p, bits, mbs, 48000, ms, trials) for p in pcms]
            for i, w in enumerate(want):
                o = int(d["data_offset"][i])
                assert bytes(img[o:o + len(w)]) == w, (seed, mapping, "encode", i, ch, bits, ms, trials, mbs, lengths[i], uniform)
            hd = parse_header(want[0][:31])
            dplan = engine.decode_plan(hd, d, True)
            d_out = torch.zeros_like(d_pcm)
            dplan.run(d_img, d_out)
            torch.cuda.synchronize()
            dplan.close()
            out = d_out.cpu().numpy()
            for i, w in enumerate(want):
                o = int(d["pcm_offset"][i])
                ref = ob.decode(w)[0].reshape(-1)
                assert np.array_equal(out[o:o + ref.size], ref), (seed, mapping, "decode", i, ch, bits, ms, trials, mbs, lengths[i], uniform)
    finally:
        engine.set_mapping("auto")
