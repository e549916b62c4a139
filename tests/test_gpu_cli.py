"""aad_batch on the GPU box: several contexts on one device (-D 0,0), waves (reader / device /
writer pipeline), list input, non-16-bit WAV input against the reference CLI's own outputs."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.synth import synth_pcm
from helpers import GOLDEN, ROOT, read_wav16, wav16_bytes, wav_bytes_depth

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "aad_amd", "aad_batch")
FIX = os.path.join(GOLDEN, "ref_fixtures")


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def test_two_contexts_on_one_device_and_waves(tmp_path):
    """`-D 0,0`: two device slots (each its own reader / device / writer threads, context, stream
    and staging) on the one GPU, inputs dealt longest-first; a tiny wave size forces many waves per
    slot.  The reference's fixtures and 60 synthetic files of mixed lengths and channel counts
    must come out exactly as the reference / the oracle write them, in both directions."""
    src, enc, dec = tmp_path / "src", tmp_path / "enc", tmp_path / "dec"
    for d in (src, enc, dec):
        d.mkdir()
    rng = np.random.default_rng(5)
    inputs = {}
    for i in range(60):
        ch = 1 if i % 3 == 0 else 2
        pcm = synth_pcm(1, int(rng.integers(1, 30000)), ch, seed=400 + i)[0]
        inputs["f%02d" % i] = pcm
        (src / ("f%02d.wav" % i)).write_bytes(wav16_bytes(pcm, 48000))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(str(src / (n + ".wav")) for n in inputs) + "\n")
    env = dict(os.environ, AAD_BATCH_WAVE_BYTES="200000")
    fixtures = [os.path.join(FIX, n + ".wav") for n in ("sin300Hz_mono", "sin300Hz")]
    subprocess.run([CLI, "-e", "-D", "0,0", "-o", str(enc), "-l", str(lst)] + fixtures, check=True, env=env, timeout=300)
    for n in ("sin300Hz_mono", "sin300Hz"):
        assert (enc / (n + ".aad")).read_bytes() == open(os.path.join(FIX, n + ".aad"), "rb").read()
    for n, pcm in inputs.items():
        assert (enc / (n + ".aad")).read_bytes() == ob.encode(pcm, 4, 1024, 48000, False, 2), n
    assert not [p for p in enc.iterdir() if p.name.endswith(".part")]  # outputs are built as <name>.part and renamed when complete
    subprocess.run([CLI, "-d", "-D", "0,0", "-o", str(dec)] + [str(p) for p in sorted(enc.iterdir())], check=True, env=env, timeout=300)
    for n in ("sin300Hz_mono", "sin300Hz"):
        assert (dec / (n + ".wav")).read_bytes() == open(os.path.join(FIX, n + "_decoded.wav"), "rb").read()
    for n, pcm in inputs.items():
        want, hd = ob.decode((enc / (n + ".aad")).read_bytes())
        assert (dec / (n + ".wav")).read_bytes() == wav16_bytes(want, 48000), n
    # -c over two slots prints one line per input, in input order
    out = subprocess.run([CLI, "-c", "-D", "0,0", "-t", "0"] + [str(src / ("f%02d.wav" % i)) for i in range(6)],
                         check=True, env=env, timeout=300, stdout=subprocess.PIPE, text=True).stdout.splitlines()
    assert len(out) == 6
    for i, line in enumerate(out):
        path, stats = line.split("\t")
        pcm = inputs["f%02d" % i]
        rec, _ = ob.decode(ob.encode(pcm, 4, 1024, 48000, False, 0))
        assert path.endswith("f%02d.wav" % i) and stats + "\n" == ob.stats_line(ob.error_stats(pcm, rec))


def test_non_16_bit_wav_input_matches_reference_cli(tmp_path):
    """8 / 24 / 32-bit PCM input (the reference keeps the top 16 bits, src/main.c:175-179): the .aad
    files must hash to what the REAL reference CLI wrote for the same WAV bytes
    (tests/golden/wav_depths.json, made by tests/golden/make_wavdepth_golden.py)."""
    cases = json.load(open(os.path.join(GOLDEN, "wav_depths.json")))["cases"]
    assert len(cases) == 24
    groups = {}
    for k, c in enumerate(cases):
        groups.setdefault(tuple(c["options"]), []).append((k, c))
    for opts, members in groups.items():  # one invocation per option set, all widths and channel counts mixed
        src, out = tmp_path / ("src_%d" % len(opts)), tmp_path / ("out_%d" % len(opts))
        src.mkdir(exist_ok=True)
        out.mkdir(exist_ok=True)
        paths = []
        for k, c in members:
            wav = wav_bytes_depth(synth_pcm(1, c["samples"], c["channels"], seed=c["seed"])[0], 48000, c["depth"], salt=c["seed"])
            assert _sha(wav) == c["wav_sha256"]
            p = src / ("case%02d.wav" % k)
            p.write_bytes(wav)
            paths.append(str(p))
        subprocess.run([CLI, "-e"] + list(opts) + ["-o", str(out)] + paths, check=True, timeout=300)
        for k, c in members:
            aad = (out / ("case%02d.aad" % k)).read_bytes()
            assert len(aad) == c["aad_bytes"] and _sha(aad) == c["aad_sha256"], c


def test_failed_wave_leaves_no_outputs_and_truncated_images_decode_to_silence(tmp_path):
    """Outputs are files created up front and filled in place.  A wave whose device call fails (one image
    ends inside a block header: the reference's INSUFFICIENT_DATA) must leave none of them behind; an
    image that merely ends early decodes like the reference's block walk and the frames it does not hold
    come out as silence."""
    src, dec = tmp_path / "src", tmp_path / "dec"
    src.mkdir()
    dec.mkdir()
    images = {}
    for i in range(5):
        pcm = synth_pcm(1, 6000 + 500 * i, 2, seed=900 + i)[0]
        images["g%d" % i] = ob.encode(pcm, 4, 1024, 48000, False, 0)
    bad = dict(images)
    bad["g3"] = images["g3"][: 31 + 1024 * 2 + 10]          # inside block 2's header
    for n, b in bad.items():
        (src / (n + ".aad")).write_bytes(b)
    r = subprocess.run([CLI, "-d", "-o", str(dec)] + [str(src / (n + ".aad")) for n in bad], timeout=300,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "failed" in r.stderr
    assert list(dec.iterdir()) == []
    ok = dict(images)
    ok["g3"] = images["g3"][: 31 + 1024 * 2 + 500]          # block 2 present in part: decoded as far as the walk goes
    for n, b in ok.items():
        (src / (n + ".aad")).write_bytes(b)
    subprocess.run([CLI, "-d", "-o", str(dec)] + [str(src / (n + ".aad")) for n in ok], check=True, timeout=300)
    for n, b in images.items():
        want, hd = ob.decode(b)
        got = read_wav16(str(dec / (n + ".wav")))[0]
        if n != "g3":
            assert (dec / (n + ".wav")).read_bytes() == wav16_bytes(want, 48000), n
        else:
            spb = 992
            assert got.shape == want.shape
            assert np.array_equal(got[: 3 * spb], ob.decode(ok["g3"])[0][: 3 * spb])
            assert not got[3 * spb:].any()


def test_reconstruct_in_place_over_the_input(tmp_path):
    """`-r -o DIR` with DIR the input's own directory: the output name is the input file.  Inputs are
    mapped files, so the output must not be created over one while it is still being read: the result is
    the reconstruction, byte for byte what `-r` into another directory gives."""
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    pcm = synth_pcm(1, 40000, 2, seed=321)[0]
    (a / "x.wav").write_bytes(wav16_bytes(pcm, 48000))
    (b / "x.wav").write_bytes(wav16_bytes(pcm, 48000))
    subprocess.run([CLI, "-r", "-o", str(a), str(a / "x.wav")], check=True, timeout=300)
    other = tmp_path / "c"
    other.mkdir()
    subprocess.run([CLI, "-r", "-o", str(other), str(b / "x.wav")], check=True, timeout=300)
    rec, _ = ob.decode(ob.encode(pcm, 4, 1024, 48000, False, 2))
    assert (other / "x.wav").read_bytes() == wav16_bytes(rec, 48000)
    assert (a / "x.wav").read_bytes() == (other / "x.wav").read_bytes()
