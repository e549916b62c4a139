"""GPU tests of the reconstruction modes: encode -> decode ->
residual / RMSE-MSD-MaxAE on the device, against the goldens made from the compiled reference CLI
(tests/golden/cli_modes.json, `aad -r` / `aad -g` / `aad -c`, src/main.c:275-503) and the oracle.

Bar: reconstructed and residual PCM bit-exact.  The three statistics are fp64 sums whose ORDER
differs from the CLI's channel-major walk by default (fixed reduction tree on the device), so they are held
to 1e-12 relative - and to the exact text of the line the CLI printed (six decimals), which the device
guarantees by taking the reference's order itself whenever the tree's result lies within the reordering bound
of a rounding boundary (aad_compare.hip.h); forced, that order gives the oracle's doubles bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.capi import make_parameter
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, cli_mode_input, sha256, wav16_bytes

pytestmark = pytest.mark.gpu
STATS_RTOL = 1e-12
CASES = json.load(open(os.path.join(GOLDEN, "cli_modes.json")))["cases"]
CLI = os.path.join(os.path.dirname(GOLDEN), "..", "aad_amd", "aad_batch")


@pytest.fixture(scope="module")
def engine():
    import torch  # noqa: F401
    from aad_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_statistics_in_the_reference_order_are_bit_identical(engine):
    """AAD_HIP_OPTION_COMPARE_ORDER = sequential: one lane per stream walks the values channel by channel, sample by
    sample, with separately rounded multiplies and adds - the reference's loop (src/main.c:478-497).  The three
    doubles must EQUAL the oracle's (==, not a tolerance): this is the path the device takes by itself whenever its
    tree sum lies close enough to a rounding boundary of the printed six decimals for the order to show, so the
    printed line is the reference's by construction.  Mono, stereo, 8 channels, lengths from one sample to several
    blocks, loud noise (large sums) and near-silence (tiny sums)."""
    engine.set_compare_order(sequential=True)
    try:
        rng = np.random.default_rng(31)
        for ch, bits, trials, ms in ((1, 4, 0, False), (2, 4, 2, False), (2, 3, 0, True), (2, 2, 1, False), (8, 3, 0, False)):
            lengths = [1, 3, 4, 5, 17, 992, 993, 2500, int(rng.integers(3000, 9000))]
            pcms = [synth_pcm(1, n, ch, seed=500 + n, kind=["music", "noise", "nyquist"][k % 3])[0] for k, n in enumerate(lengths)]
            pcms.append((synth_pcm(1, 4000, ch, seed=9)[0] // 4096).astype(np.int16))  # a few LSBs of signal
            param = make_parameter(ch, bits, 1024, 48000, ms, trials)
            rec, stats = engine.reconstruct_host(pcms, param, residual=False)
            for i, pcm in enumerate(pcms):
                want = ob.error_stats(pcm, rec[i])
                assert _as_tuple(stats[i]) == want, (ch, bits, trials, ms, len(pcm), _as_tuple(stats[i]), want)
    finally:
        engine.set_compare_order(sequential=False)
    # and the default order prints the same line for the same inputs
    for ch, bits in ((1, 4), (2, 2)):
        pcms = [synth_pcm(1, n, ch, seed=600 + n)[0] for n in (5, 992, 4001)]
        param = make_parameter(ch, bits, 1024, 48000, False, 0)
        rec, stats = engine.reconstruct_host(pcms, param, residual=False)
        for i, pcm in enumerate(pcms):
            assert ob.stats_line(_as_tuple(stats[i])) == ob.stats_line(ob.error_stats(pcm, rec[i]))


def _as_tuple(rec):
    return float(rec["rms_error"]), float(rec["mean_abs_error"]), float(rec["max_abs_error"])


def test_reconstruct_batch_matches_reference_cli(engine):
    """every golden case through AADHip_ReconstructBatch, grouped per parameter set so that mono and
    stereo inputs of different lengths share one call"""
    groups = {}
    for c in CASES:
        pcm, rate, _ = cli_mode_input(c)
        groups.setdefault((pcm.shape[1], rate, c["bits"], c["trials"], c["ms"]), []).append((c, pcm))
    assert len(groups) >= 12
    for (ch, rate, bits, trials, ms), members in groups.items():
        param = make_parameter(ch, bits, 1024, rate, ms, trials)
        pcms = [m[1] for m in members]
        rec, stats = engine.reconstruct_host(pcms, param, residual=False)
        gap, stats2 = engine.reconstruct_host(pcms, param, residual=True)
        only_stats = engine.reconstruct_host(pcms, param, want_pcm=False)[1]
        for i, (c, pcm) in enumerate(members):
            assert sha256(wav16_bytes(rec[i], rate)) == c["reconstructed_sha256"], c
            assert sha256(wav16_bytes(gap[i], rate)) == c["residual_sha256"], c
            assert ob.stats_line(_as_tuple(stats[i])) == c["stats_line"], c
            want = ob.error_stats(pcm, rec[i])
            np.testing.assert_allclose(_as_tuple(stats[i]), want, rtol=STATS_RTOL, atol=0)
            assert _as_tuple(stats2[i]) == _as_tuple(stats[i]) == _as_tuple(only_stats[i])


@pytest.mark.parametrize("tile_kbytes", [1, 7, 300])
def test_reconstruct_batch_in_chunks_and_waves(engine, tile_kbytes):
    """AADHip_ReconstructBatch stages its PCM through the pinned blocks in chunks of the tile budget and, when the batch does
    not fit the device, runs it as several waves of whole streams (aad_hip_engine.hip, reconstruct_wave): a forced tile size
    (AAD_HIP_OPTION_TILE_KBYTES) makes the chunks a few KiB and the waves 64 tiles' worth, so that ragged batches - one-sample
    streams next to streams longer than a chunk AND longer than a wave, 1 / 2 / 8 channels, trial search, M/S - cross every
    boundary: chunk edges inside a stream's row, rows inside the padding between streams, the statistics block split over
    chunks.  Same bytes and the same doubles as the call with everything in one piece; PCM and the printed line == the oracle."""
    rng = np.random.default_rng(900 + tile_kbytes)
    for ch, bits, trials, ms, count in ((2, 4, 2, False, 90), (1, 3, 0, False, 400), (8, 2, 0, False, 40), (2, 2, 1, True, 130)):
        lengths = [int(rng.choice([1, 3, 4, 5, 200, 992, 993, int(rng.integers(1, 5000)), int(rng.integers(5000, 40000))])) for _ in range(count)]
        lengths[count // 2] = 150000  # longer than any wave at tile_kbytes = 1 (64 KiB): a wave of its own
        pcms = [synth_pcm(1, n, ch, seed=int(rng.integers(0, 1 << 30)), kind=str(rng.choice(["music", "noise"])))[0] for n in lengths]
        param = make_parameter(ch, bits, 1024, 48000, ms, trials)
        engine.set_tile_kbytes(0)
        rec0, stats0 = engine.reconstruct_host(pcms, param, residual=False)
        gap0, _ = engine.reconstruct_host(pcms, param, residual=True)
        engine.set_tile_kbytes(tile_kbytes)
        try:
            rec, stats = engine.reconstruct_host(pcms, param, residual=False)
            gap, stats_g = engine.reconstruct_host(pcms, param, residual=True)
            only = engine.reconstruct_host(pcms, param, want_pcm=False)[1]
            no_stats = engine.reconstruct_host(pcms, param, want_stats=False)[0]
        finally:
            engine.set_tile_kbytes(0)
        for i, pcm in enumerate(pcms):
            assert np.array_equal(rec[i], rec0[i]) and np.array_equal(gap[i], gap0[i]) and np.array_equal(no_stats[i], rec0[i]), (tile_kbytes, ch, bits, i, lengths[i])
            assert _as_tuple(stats[i]) == _as_tuple(stats0[i]) == _as_tuple(stats_g[i]) == _as_tuple(only[i]), (tile_kbytes, ch, bits, i, lengths[i])
        for i in list(range(0, count, 9)) + [count // 2]:
            want = ob.decode(ob.encode(pcms[i], bits, 1024, 48000, ms, trials))[0]
            assert np.array_equal(rec[i], want) and np.array_equal(gap[i], ob.residual(pcms[i], want)), (tile_kbytes, ch, bits, i, lengths[i])
            assert ob.stats_line(_as_tuple(stats[i])) == ob.stats_line(ob.error_stats(pcms[i], want))


@pytest.mark.parametrize("streams,samples,ch,bits", [(1000, 992, 2, 4), (3, 70001, 1, 3), (64, 4000, 8, 2)])
def test_reconstruct_device_resident_vs_oracle(engine, streams, samples, ch, bits):
    """device-resident form (AADHip_ReconstructPlanRun): nothing but the statistics needs to leave
    HBM; checked against oracle encode -> decode -> residual / statistics on sampled streams,
    including the 8-channel container extension and streams longer than one compare segment"""
    import torch
    pcm = synth_pcm(streams, samples, ch, seed=99)
    param = make_parameter(ch, bits, 1024, 48000, False, 0)
    d_pcm = torch.from_numpy(pcm).cuda()
    rec, stats = engine.reconstruct_uniform(d_pcm, param, residual=False)
    gap, stats_g = engine.reconstruct_uniform(d_pcm, param, residual=True)
    torch.cuda.synchronize()
    rec, gap, stats, stats_g = rec.cpu().numpy(), gap.cpu().numpy(), stats.cpu().numpy(), stats_g.cpu().numpy()
    assert np.array_equal(stats, stats_g)
    for s in range(0, streams, max(1, streams // 7)):
        want = ob.decode(ob.encode(pcm[s], bits, 1024))[0]
        assert np.array_equal(rec[s], want), s
        assert np.array_equal(gap[s], ob.residual(pcm[s], want)), s
        np.testing.assert_allclose(stats[s], ob.error_stats(pcm[s], want), rtol=STATS_RTOL, atol=0)


def test_reconstruct_wraparound_corner(engine):
    """|x - y| >= 32768 cannot come out of the codec for real input, but the residual's int16 wrap
    and the statistics' 32-bit wrap (src/main.c:419-423, 470-474) are part of what the CLI does:
    a full-scale square wave at 2 bits drives the reconstruction far enough off to exercise large
    differences; the oracle restates the same wraps"""
    n = 3000
    x = np.where((np.arange(n) // 3) % 2 == 0, 32767, -32768).astype(np.int16).reshape(-1, 1)
    param = make_parameter(1, 2, 1024, 48000, False, 0)
    rec, stats = engine.reconstruct_host([x], param)
    gap, _ = engine.reconstruct_host([x], param, residual=True)
    want = ob.decode(ob.encode(x, 2, 1024))[0]
    assert np.array_equal(rec[0], want)
    assert np.array_equal(gap[0], ob.residual(x, want))
    np.testing.assert_allclose(_as_tuple(stats[0]), ob.error_stats(x, want), rtol=STATS_RTOL, atol=0)


def test_reconstruct_argument_errors(engine):
    import ctypes as C
    lib, ctx = engine.lib, engine._ctx
    param = make_parameter(2, 4, 1024, 48000, False, 0)
    x = np.zeros((10, 2), dtype=np.int16)
    n = (C.c_uint32 * 1)(10)
    pp = (C.c_void_p * 1)(x.ctypes.data)
    assert lib.AADHip_ReconstructBatch(None, C.byref(param), 1, pp, n, 0, None, None) == 1
    assert lib.AADHip_ReconstructBatch(ctx, C.byref(param), 1, None, n, 0, None, None) == 1
    assert lib.AADHip_ReconstructBatch(ctx, C.byref(param), 1, pp, n, 7, pp, None) == 1          # unknown output kind
    bad = make_parameter(2, 5, 1024, 48000, False, 0)
    assert lib.AADHip_ReconstructBatch(ctx, C.byref(bad), 1, pp, n, 0, None, None) == 2          # INVALID_FORMAT
    assert lib.AADHip_ReconstructBatch(ctx, C.byref(param), 0, None, None, 0, None, None) == 0   # empty batch


def test_aad_batch_cli_reconstruction_modes(tmp_path):
    """aad_batch -r / -g / -c over ALL golden inputs in one invocation each (mono and stereo mixed),
    with two worker threads (-D 0,0: two contexts on the one GPU of the test box - the code path
    `-D 0,1,...` takes on a multi-GPU node) and the inputs given through a list file"""
    assert os.path.exists(CLI), "aad_batch not built"
    names, wavs = [], {}
    for c in CASES:
        if c["input"] not in wavs:
            _, _, data = cli_mode_input(c)
            path = tmp_path / (c["input"] + ".wav")
            path.write_bytes(data)
            wavs[c["input"]] = str(path)
            names.append(c["input"])
    listfile = tmp_path / "inputs.txt"
    listfile.write_text("\n".join(wavs[n] for n in names[1:]) + "\n")
    for bits, trials, ms in ((4, 2, False), (3, 0, True), (2, 2, False)):
        # the reference rejects M/S on mono input, and so does the engine: stereo inputs only there
        use = [n for n in names if not (ms and "mono" in n or ms and "_c1_" in n)]
        opts = ["-b", str(bits), "-t", str(trials), "-D", "0,0"] + (["-m"] if ms else [])
        want = {(c["input"]): c for c in CASES if (c["bits"], c["trials"], c["ms"]) == (bits, trials, ms)}
        rdir, gdir = tmp_path / ("r%d%d%d" % (bits, trials, ms)), tmp_path / ("g%d%d%d" % (bits, trials, ms))
        rdir.mkdir()
        gdir.mkdir()
        if use == names:
            files = ["-l", str(listfile), wavs[names[0]]]
        else:
            files = [wavs[n] for n in use]
        # the third parameter set with a forced tile size: the library's chunked staging and several device waves under the CLI
        env = dict(os.environ, AAD_HIP_TILE_KBYTES="3") if bits == 2 else None
        subprocess.run([CLI, "-r"] + opts + ["-o", str(rdir)] + files, check=True, env=env)
        subprocess.run([CLI, "--gap"] + opts + ["--output-dir", str(gdir)] + files, check=True, env=env)
        out = subprocess.run([CLI, "-c"] + opts + files, check=True, capture_output=True, text=True, env=env).stdout
        lines = dict(l.split("\t", 1) for l in out.splitlines(keepends=True))
        assert len(lines) == len(use)
        for n in use:
            assert sha256((rdir / (n + ".wav")).read_bytes()) == want[n]["reconstructed_sha256"], (n, bits)
            assert sha256((gdir / (n + ".wav")).read_bytes()) == want[n]["residual_sha256"], (n, bits)
            assert lines[wavs[n]] == want[n]["stats_line"], (n, bits)
