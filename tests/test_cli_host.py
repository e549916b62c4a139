"""aad_batch (aad_amd/cli/aad_batch.c): what it must refuse BEFORE any device work - runs without a GPU."""
import os
import subprocess

import numpy as np

from aad_amd.synth import synth_pcm
from helpers import ROOT, wav16_bytes, wav_bytes_depth

CLI = os.path.join(ROOT, "aad_amd", "aad_batch")


def _run(args, **kw):
    return subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60, **kw)


def test_information_help_and_version_match_the_reference_cli(tmp_path):
    """-i / -h / -v (reference src/main.c:229-272, 505-547): the text the compiled reference CLI printed
    (tests/golden/cli_info.json, make_cli_info_golden.py), for images rebuilt here with the oracle.  No device work."""
    import hashlib
    import json
    import oracle_binding as ob
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cli_info.json")))
    paths = []
    for k, c in enumerate(gold["cases"]):
        img = ob.encode(synth_pcm(1, c["samples"], c["channels"], seed=c["seed"])[0], c["bits"], c["max_block_size"], c["rate"], c["ms"], 0)
        assert hashlib.sha256(img).hexdigest() == c["image_sha256"]  # the bytes the reference CLI wrote
        p = tmp_path / ("case%d.aad" % k)
        p.write_bytes(img)
        paths.append(str(p))
        r = _run(["-i", str(p)])
        assert r.returncode == 0 and r.stdout == c["information"], (k, r.stdout, r.stderr)
        assert _run(["--information", str(p)]).stdout == c["information"]
    # several inputs: every report behind a line that names the file
    r = _run(["-i"] + paths)
    assert r.returncode == 0 and r.stdout == "".join(p + "\n" + c["information"] for p, c in zip(paths, gold["cases"]))
    # errors as the reference words them (src/main.c:240-258), the other files are still reported
    bad = tmp_path / "short.aad"
    bad.write_bytes(b"AAD\0\0\0")
    r = _run(["-i", str(bad), paths[0]])
    assert r.returncode == 1 and "Failed to read from" in r.stderr and gold["cases"][0]["information"] in r.stdout
    junk = tmp_path / "junk.aad"
    junk.write_bytes(b"RIFF" + bytes(40))
    r = _run(["-i", str(junk)])
    assert r.returncode == 1 and "Failed to read header. API result:" in r.stderr
    assert _run(["-i", str(tmp_path / "missing.aad")]).returncode == 1
    # -v: the reference's line; -h: its option table line for line, this front end's own options behind it
    for flag in ("-v", "--version"):
        r = _run([flag])
        assert r.returncode == 0 and r.stdout == gold["version"]
    r = _run(["-h"])
    assert r.returncode == 0 and "options: \n" in r.stdout
    lines = [l for l in r.stdout.splitlines(True) if l.startswith("  -")]
    assert lines[:len(gold["help_option_lines"])] == gold["help_option_lines"]
    assert any("--devices" in l for l in lines) and any("--list" in l for l in lines) and any("--output-dir" in l for l in lines)
    assert _run(["-e", "-h", "x.wav"]).returncode == 0  # help wins over a mode, as in the reference


def test_usage_errors():
    assert _run([]).returncode == 2
    assert _run(["-e", "x.wav"]).returncode == 2          # no output directory
    assert _run(["-e", "-o", "/tmp", "-D", "zero", "x.wav"]).returncode == 2
    assert _run(["-q", "-o", "/tmp", "x.wav"]).returncode == 2


def test_same_stem_inputs_are_refused(tmp_path):
    """a/track.wav and b/track.wav would both become OUTDIR/track.aad (advisor finding, round 1)"""
    wav = wav16_bytes(synth_pcm(1, 500, 2, seed=3)[0], 48000)
    for d in ("a", "b"):
        (tmp_path / d).mkdir()
        (tmp_path / d / "track.wav").write_bytes(wav)
    (tmp_path / "out").mkdir()
    r = _run(["-e", "-o", str(tmp_path / "out"), str(tmp_path / "a" / "track.wav"), str(tmp_path / "b" / "track.wav")])
    assert r.returncode == 1 and "would write the same output file" in r.stderr
    assert not list((tmp_path / "out").iterdir())
    # different extensions, same stem: still one output name
    (tmp_path / "a" / "track.wave").write_bytes(wav)
    r = _run(["-e", "-o", str(tmp_path / "out"), str(tmp_path / "a" / "track.wav"), str(tmp_path / "a" / "track.wave")])
    assert r.returncode == 1 and "would write the same output file" in r.stderr


def test_list_file_with_bare_cr_line_endings(tmp_path):
    """A list with CR-only endings holds more paths than it has '\\n' bytes (round 1 sized its array
    by counting '\\n': heap overflow).  All 300 entries must be seen: the first missing file is named."""
    names = ["missing_%03d.wav" % i for i in range(300)]
    lst = tmp_path / "list.txt"
    lst.write_bytes("\r".join(str(tmp_path / n) for n in names).encode() + b"\r")
    r = _run(["-c", "-l", str(lst)])
    assert r.returncode == 1 and "cannot read" in r.stderr and "missing_" in r.stderr
    # mixed endings and blank lines
    lst.write_bytes(("\r\n".join(str(tmp_path / n) for n in names[:5]) + "\n\n\r" + str(tmp_path / "last.wav")).encode())
    r = _run(["-c", "-l", str(lst)])
    assert r.returncode == 1 and "cannot read" in r.stderr


def test_wav_depth_helper_and_converter_agree():
    """AADWav_ConvertToPcm16 (the reference's top-16-bit rule, src/main.c:175-179 + src/wav.c:392-417)
    on the helper's 8/16/24/32-bit images gives back the int16 the image was built from."""
    import ctypes as C
    import aad_amd
    from aad_amd.capi import AADWavInfo
    lib = aad_amd.load_library()
    pcm = synth_pcm(1, 777, 2, seed=21, kind="noise")[0]
    for depth in (8, 16, 24, 32):
        wav = np.frombuffer(wav_bytes_depth(pcm, 44100, depth, salt=5), dtype=np.uint8)
        info = AADWavInfo()
        assert lib.AADWav_ParseHeader(wav.ctypes.data, len(wav), C.byref(info)) == 0
        assert (info.bits_per_sample, info.num_channels, info.num_samples, info.sampling_rate) == (depth, 2, 777, 44100)
        out = np.zeros((777, 2), dtype=np.int16)
        assert lib.AADWav_ConvertToPcm16(wav.ctypes.data + info.data_offset, depth, 777 * 2, out.ctypes.data) == 0
        want = pcm if depth != 8 else ((pcm.astype(np.int32) >> 8) << 8).astype(np.int16)
        assert np.array_equal(out, want), depth
    assert lib.AADWav_ConvertToPcm16(wav.ctypes.data, 12, 4, out.ctypes.data) == 2  # INVALID_FORMAT
