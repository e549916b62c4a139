#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container: it needs oracle/_ref/libaadref.so, which oracle/Makefile
compiles from the reference's own sources where they lie under /root/reference (nothing of the
reference is copied into this repo - only input/output DATA is kept):

  ref_fixtures/   the reference's own test data files (test/sin300Hz*.wav|aad|_decoded.wav,
                  test/unit_impulse*.wav) - fixtures its test-suite pins decode with
                  (test/test_aad_decoder.c:256-339)
  manifest.json   for a parameter matrix over the deterministic synthetic corpus
                  (aad_amd/synth.py): SHA-256 of the reference's .aad image and of its decode
  cases/*.aad     full bytes of a handful of small cases, for debugging a hash mismatch

The GPU box has no reference; tests/test_gpu_golden.py rebuilds the same inputs there and
compares the HIP engine's bytes with these hashes.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import aad_amd  # noqa: E402
from aad_amd.synth import synth_pcm  # noqa: E402

REF_TEST_DIR = "/root/reference/test"
REF_FILES = ["sin300Hz.wav", "sin300Hz_mono.wav", "sin300Hz.aad", "sin300Hz_mono.aad",
             "sin300Hz_decoded.wav", "sin300Hz_mono_decoded.wav", "unit_impulse.wav", "unit_impulse_mono.wav"]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def case_matrix():
    """(name, kind, channels, bits, ms, trials, max_block_size, samples, seed)"""
    cases = []
    for kind in ("music", "noise", "nyquist"):
        for ch in (1, 2):
            for bits in (4, 3, 2):
                for ms in ((False, True) if ch == 2 else (False,)):
                    for trials in (0, 1, 2):
                        for mbs, n in ((128, 777), (1024, 2), (1024, 4), (1024, 5), (1024, 3001), (4096, 9000)):
                            if kind != "music" and (mbs, n) not in ((128, 777), (1024, 3001)):
                                continue
                            name = "%s_c%d_b%d_%s_t%d_s%d_n%d" % (kind, ch, bits, "ms" if ms else "lr", trials, mbs, n)
                            cases.append((name, kind, ch, bits, ms, trials, mbs, n, 1234))
    return cases


def main():
    ref = aad_amd.LegacyCodec(aad_amd.load_library(os.path.join(ROOT, "oracle", "_ref", "libaadref.so"), hip=False))
    os.makedirs(os.path.join(HERE, "ref_fixtures"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "cases"), exist_ok=True)
    for f in REF_FILES:
        shutil.copyfile(os.path.join(REF_TEST_DIR, f), os.path.join(HERE, "ref_fixtures", f))

    manifest = {"generator": "tests/golden/make_golden.py", "reference": "aikiriao/AAD codec v18 / format v4",
                "cases": []}
    for name, kind, ch, bits, ms, trials, mbs, n, seed in case_matrix():
        pcm = synth_pcm(1, n, ch, seed=seed, kind=kind)[0]
        aad = ref.encode(pcm, bits, mbs, 48000, ms, trials)
        dec, _ = ref.decode(aad)
        entry = dict(name=name, kind=kind, channels=ch, bits=bits, ms=ms, trials=trials, max_block_size=mbs,
                     samples=n, seed=seed, pcm_sha256=sha(pcm.tobytes()), aad_bytes=len(aad), aad_sha256=sha(aad),
                     decoded_sha256=sha(dec.tobytes()))
        if len(aad) <= 2048 and kind == "music" and trials in (0, 2):
            with open(os.path.join(HERE, "cases", name + ".aad"), "wb") as f:
                f.write(aad)
            entry["file"] = "cases/%s.aad" % name
        manifest["cases"].append(entry)

    # BASELINE config corpora, hashed per batch (the reference encodes stream by stream)
    corpora = []
    for cname, streams, n, ch, bits, trials in (("cfg2_stereo4_1000x1blk_t0", 1000, 992, 2, 4, 0),
                                                ("cfg2_stereo4_1000x1blk_t2", 1000, 992, 2, 4, 2),
                                                ("cfg4ref_stereo3_2000x1blk_t0", 2000, 1316, 2, 3, 0),
                                                ("cfg4ref_stereo2_2000x1blk_t0", 2000, 1980, 2, 2, 0),
                                                ("cfg5_stereo4_100x10blk_t0", 100, 9920, 2, 4, 0)):
        pcm = synth_pcm(streams, n, ch, seed=1234)
        h_aad, h_dec = hashlib.sha256(), hashlib.sha256()
        size = 0
        for s in range(streams):
            aad = ref.encode(pcm[s], bits, 1024, 48000, False, trials)
            dec, _ = ref.decode(aad)
            h_aad.update(aad)
            h_dec.update(dec.tobytes())
            size = len(aad)
        corpora.append(dict(name=cname, streams=streams, samples=n, channels=ch, bits=bits, trials=trials,
                            max_block_size=1024, seed=1234, image_bytes=size, pcm_sha256=sha(pcm.tobytes()),
                            aad_concat_sha256=h_aad.hexdigest(), decoded_concat_sha256=h_dec.hexdigest()))
    manifest["corpora"] = corpora

    # 8-channel container extension: the reference cannot produce it, so each
    # channel is pinned as a mono stream whose block geometry equals the 8-channel one.
    eight = []
    for bits, mono_mbs in ((4, 128), (3, 128), (2, 128)):
        pcm = synth_pcm(4, 1000, 8, seed=77)
        for s in range(4):
            for c in range(8):
                aad = ref.encode(pcm[s][:, c:c + 1], bits, mono_mbs, 48000, False, 0)
                eight.append(dict(stream=s, channel=c, bits=bits, mono_max_block_size=mono_mbs, samples=1000,
                                  seed=77, aad_sha256=sha(aad)))
    manifest["eight_channel_as_mono"] = eight

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("cases:", len(manifest["cases"]), "corpora:", len(corpora), "8ch lanes:", len(eight))


if __name__ == "__main__":
    main()
