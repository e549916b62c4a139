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
                                                ("cfg5_stereo4_100x10blk_t0", 100, 9920, 2, 4, 0),
                                                # long in-lane block chains (src/aad_encoder.c:853-886, header carry :645-653)
                                                ("cfg2iii_stereo4_1x1000blk_t0", 1, 992000, 2, 4, 0),
                                                ("cfg2iii_stereo4_1x1000blk_t2", 1, 992000, 2, 4, 2),
                                                ("cfg2ii_stereo4_1000x16blk_t0", 1000, 15872, 2, 4, 0),
                                                ("cfg5_stereo4_1250x10blk_t0", 1250, 9920, 2, 4, 0),
                                                ("chain_stereo4_40x25blk_t2", 40, 24800 - 77, 2, 4, 2),
                                                ("chain_mono2_200x5blk_t0", 200, 20140, 1, 2, 0),
                                                ("chain_mono2_60x5blk_t2", 60, 20140 - 1001, 1, 2, 2),
                                                ("chain_mono3_100x4blk_t1", 100, 4 * 2684 - 100, 1, 3, 1)):
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

    # BASELINE config 5 at full size (10 000 stereo 4-bit files x 10 blocks), for bench.py's batched-file
    # leg at 1 / 2 / 4 / 8 ranks x 1250 files: SHA-256 of the images of the first N files, in job order
    pcm_hash, h = hashlib.sha256(), hashlib.sha256()
    prefix = {}
    for s in range(10000):
        pcm = synth_pcm(1, 9920, 2, seed=1234, first_stream=s)[0]
        pcm_hash.update(pcm.tobytes())
        h.update(ref.encode(pcm, 4, 1024, 48000, False, 0))
        if s + 1 in (1250, 2500, 5000, 10000):
            prefix[str(s + 1)] = h.copy().hexdigest()
    manifest["file_corpora"] = [dict(name="cfg5_stereo4_10000x10blk_t0", streams=10000, samples=9920, channels=2, bits=4,
                                     trials=0, max_block_size=1024, seed=1234, image_bytes=10271,
                                     pcm_sha256=pcm_hash.hexdigest(), aad_prefix_sha256=prefix)]

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

    # BASELINE config 4 at full size: 10 000 eight-channel one-block segments, 3- and 2-bit, pinned
    # the same way - every (segment, channel) as the reference's mono image, hashed in that order
    eight_corpora = []
    for bits, spb, block_size, mono_bs in ((3, 292, 1008, 126), (2, 444, 1024, 128), (4, 224, 1024, 128)):
        streams = 10000 if bits != 4 else 1000
        pcm = synth_pcm(streams, spb, 8, seed=1234)
        h = hashlib.sha256()
        size = 0
        for s in range(streams):
            for c in range(8):
                aad = ref.encode(pcm[s][:, c:c + 1], bits, 128, 48000, False, 0)
                h.update(aad)
                size = len(aad)
        eight_corpora.append(dict(name="cfg4_8ch_b%d_%dx1blk_t0" % (bits, streams), streams=streams, samples=spb,
                                  channels=8, bits=bits, trials=0, max_block_size=1024, block_size=block_size,
                                  mono_max_block_size=128, mono_block_size=mono_bs, mono_image_bytes=size, seed=1234,
                                  pcm_sha256=sha(pcm.tobytes()), mono_images_concat_sha256=h.hexdigest()))
    manifest["eight_channel_corpora"] = eight_corpora

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("cases:", len(manifest["cases"]), "corpora:", len(corpora), "8ch lanes:", len(eight),
          "8ch corpora:", len(eight_corpora))


if __name__ == "__main__":
    main()
