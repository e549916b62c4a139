#!/usr/bin/env python3
"""Golden outputs of the reference CLI's reconstruction modes (-r, -g, -c; src/main.c:275-503).

Runs only in the build container: it executes oracle/_ref/aad, the reference CLI compiled by
oracle/Makefile from the reference's own sources where they lie under /root/reference.  Inputs
are the reference's fixture WAVs (already under ref_fixtures/) and synthetic WAVs rebuilt from
aad_amd/synth.py; what is kept in cli_modes.json is DATA only: the text `aad -c` printed and the
SHA-256 of the WAV files `aad -r` / `aad -g` wrote.
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from aad_amd.synth import synth_pcm  # noqa: E402
from helpers import sha256, wav16_bytes  # noqa: E402

CLI = os.path.join(ROOT, "oracle", "_ref", "aad")


def inputs():
    """(name, source) - source is a fixture file name or a synth recipe"""
    out = [("sin300Hz", dict(fixture="sin300Hz.wav")), ("sin300Hz_mono", dict(fixture="sin300Hz_mono.wav")),
           ("unit_impulse", dict(fixture="unit_impulse.wav"))]
    for kind, ch, n in (("music", 2, 5000), ("music", 1, 3001), ("noise", 2, 4000), ("nyquist", 2, 2500)):
        out.append(("%s_c%d_n%d" % (kind, ch, n), dict(kind=kind, channels=ch, samples=n, seed=4321)))
    return out


def main():
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, src in inputs():
            wav = os.path.join(tmp, name + ".wav")
            if "fixture" in src:
                data = open(os.path.join(HERE, "ref_fixtures", src["fixture"]), "rb").read()
            else:
                data = wav16_bytes(synth_pcm(1, src["samples"], src["channels"], seed=src["seed"], kind=src["kind"])[0], 48000)
            open(wav, "wb").write(data)
            for bits in (4, 3, 2):
                for trials in (0, 2):
                    for ms in ((False, True) if name not in ("sin300Hz_mono", "music_c1_n3001") else (False,)):
                        opts = ["-b", str(bits), "-s", "1024", "-t", str(trials)] + (["-m"] if ms else [])
                        line = subprocess.run([CLI, "-c"] + opts + [wav], check=True, capture_output=True, text=True).stdout
                        rec, gap = os.path.join(tmp, "r.wav"), os.path.join(tmp, "g.wav")
                        subprocess.run([CLI, "-r"] + opts + [wav, rec], check=True, capture_output=True)
                        subprocess.run([CLI, "-g"] + opts + [wav, gap], check=True, capture_output=True)
                        cases.append(dict(input=name, source=src, bits=bits, trials=trials, ms=ms,
                                          input_sha256=sha256(data), stats_line=line,
                                          reconstructed_sha256=sha256(open(rec, "rb").read()),
                                          residual_sha256=sha256(open(gap, "rb").read())))
    with open(os.path.join(HERE, "cli_modes.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_cli_golden.py", cases=cases), f, indent=1)
    print("cli mode cases:", len(cases))


if __name__ == "__main__":
    main()
