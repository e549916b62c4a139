#!/usr/bin/env python3
"""Golden vectors for non-16-bit WAV input (the reference CLI accepts 8/16/24/32-bit PCM and keeps the
top 16 bits of every sample, src/main.c:175-179 with src/wav.c:392-417): run the REAL reference
CLI (oracle/_ref/aad, compiled by oracle/Makefile from the sources under /root/reference) on WAV
files of each width built from the deterministic corpus and record the SHA-256 of the .aad it
writes.  Build container only; tests/test_gpu_cli.py rebuilds the same WAV bytes on the GPU box
(helpers.wav_bytes_depth) and holds aad_batch to these hashes."""
import hashlib
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
from helpers import wav_bytes_depth  # noqa: E402

REF_CLI = os.path.join(ROOT, "oracle", "_ref", "aad")


def main():
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for depth in (8, 16, 24, 32):
            for ch in (1, 2):
                for n, seed, opts in ((3001, 11, []), (9000, 12, ["-b", "3", "-t", "0"]), (2500, 13, ["-b", "2", "-s", "256", "-t", "1"])):
                    if ch == 2 and opts:
                        opts = opts + ["-m"]
                    pcm = synth_pcm(1, n, ch, seed=seed)[0]
                    wav = wav_bytes_depth(pcm, 48000, depth, salt=seed)
                    src, dst = os.path.join(tmp, "in.wav"), os.path.join(tmp, "out.aad")
                    open(src, "wb").write(wav)
                    subprocess.run([REF_CLI, "-e"] + opts + [src, dst], check=True, stdout=subprocess.DEVNULL)
                    aad = open(dst, "rb").read()
                    cases.append(dict(depth=depth, channels=ch, samples=n, seed=seed, options=opts,
                                      wav_sha256=hashlib.sha256(wav).hexdigest(), aad_bytes=len(aad),
                                      aad_sha256=hashlib.sha256(aad).hexdigest()))
    json.dump({"generator": "tests/golden/make_wavdepth_golden.py", "reference_cli": "oracle/_ref/aad -e", "cases": cases},
              open(os.path.join(HERE, "wav_depths.json"), "w"), indent=1)
    print("cases:", len(cases))


if __name__ == "__main__":
    main()
