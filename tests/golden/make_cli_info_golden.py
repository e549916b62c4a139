#!/usr/bin/env python3
"""Golden text of the reference CLI's -i (header information), -h and -v (src/main.c:229-272, 505-547).

Runs only in the build container: it executes oracle/_ref/aad, the reference CLI compiled by oracle/Makefile
from the reference's own sources where they lie under /root/reference.  The .aad inputs are made by that same
CLI (`aad -e`) from synthetic WAVs rebuilt from aad_amd/synth.py; what is kept in cli_info.json is DATA only:
the encode parameters of each case, the SHA-256 of the image (so that the test can check it rebuilt the same
bytes with the oracle) and the text the reference printed.
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


def main():
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for ch, n, rate, bits, mbs, ms in ((2, 5000, 48000, 4, 1024, False), (1, 3001, 48000, 4, 1024, False), (2, 2500, 44100, 3, 1024, True),
                                           (1, 4000, 8000, 2, 256, False), (2, 999, 96000, 2, 2048, True), (1, 17, 22050, 3, 300, False)):
            wav, aad = os.path.join(tmp, "x.wav"), os.path.join(tmp, "x.aad")
            open(wav, "wb").write(wav16_bytes(synth_pcm(1, n, ch, seed=77)[0], rate))
            opts = ["-b", str(bits), "-s", str(mbs), "-t", "0"] + (["-m"] if ms else [])
            subprocess.run([CLI, "-e"] + opts + [wav, aad], check=True, capture_output=True)
            text = subprocess.run([CLI, "-i", aad], check=True, capture_output=True, text=True).stdout
            cases.append(dict(channels=ch, samples=n, rate=rate, bits=bits, max_block_size=mbs, ms=ms, seed=77,
                              image_sha256=sha256(open(aad, "rb").read()), information=text))
        version = subprocess.run([CLI, "-v"], check=True, capture_output=True, text=True).stdout
        help_text = subprocess.run([CLI, "-h"], check=True, capture_output=True, text=True).stdout
    option_lines = [l for l in help_text.splitlines(True) if l.startswith("  -")]
    with open(os.path.join(HERE, "cli_info.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_cli_info_golden.py", cases=cases, version=version, help_option_lines=option_lines), f, indent=1)
    print("cli info cases:", len(cases), "option lines:", len(option_lines))


if __name__ == "__main__":
    main()
