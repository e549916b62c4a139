"""aad_batch (aad_amd/cli/aad_batch.c): what it must refuse BEFORE any device work - runs without a GPU."""
import os
import subprocess

import numpy as np

from aad_amd.synth import synth_pcm
from helpers import ROOT, wav16_bytes, wav_bytes_depth

CLI = os.path.join(ROOT, "aad_amd", "aad_batch")


def _run(args, **kw):
    return subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60, **kw)


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
