"""Measurement aid: wall-clock time of the batch CLI (aad_amd/aad_batch) against the reference CLI
(oracle/_ref/aad, built in the build container from the reference's own sources) on the same files:
FILES stereo 48 kHz 16-bit WAVs of SECONDS seconds each in a tmpfs directory, encode then decode,
reference defaults (4-bit, 1024-byte blocks, trials 2 unless given).  The reference runs one process
per file - sequentially on one core, and 16 at a time - which is how its CLI is used.
usage: python tools/cli_bench.py [files=64] [seconds=30] [trials=2]     (prints one JSON object)"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aad_amd.synth import synth_pcm  # noqa: E402
from helpers import wav16_bytes  # noqa: E402

files = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 2
frames = int(48000 * seconds)
base = "/dev/shm/aad_cli_bench"
shutil.rmtree(base, ignore_errors=True)
for d in ("src", "gpu_enc", "gpu_dec", "ref_enc", "ref_dec"):
    os.makedirs(os.path.join(base, d))
pcm = synth_pcm(files, frames, 2, seed=77)
names = ["s%04d" % i for i in range(files)]
for i, n in enumerate(names):
    with open(os.path.join(base, "src", n + ".wav"), "wb") as f:
        f.write(wav16_bytes(pcm[i], 48000))
del pcm
cli, ref = os.path.join(ROOT, "aad_amd", "aad_batch"), os.path.join(ROOT, "oracle", "_ref", "aad")
src = [os.path.join(base, "src", n + ".wav") for n in names]


def timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def sha_dir(d):
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        with open(os.path.join(d, n), "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


out = {"files": files, "seconds_each": seconds, "trials": trials, "frames_per_file": frames,
       "msamples_total": round(files * frames * 2 / 1e6, 1)}
# the batch CLI: one process, all files (a first tiny run pays process start and context creation apart)
subprocess.run([cli, "-e", "-t", str(trials), "-o", os.path.join(base, "gpu_enc"), src[0]], check=True)
out["batch_encode_s"] = round(timed(lambda: subprocess.run([cli, "-e", "-t", str(trials), "-o", os.path.join(base, "gpu_enc")] + src, check=True)), 3)
aads = [os.path.join(base, "gpu_enc", n + ".aad") for n in names]
out["batch_decode_s"] = round(timed(lambda: subprocess.run([cli, "-d", "-o", os.path.join(base, "gpu_dec")] + aads, check=True)), 3)


def ref_encode(n):
    subprocess.run([ref, "-e", "-t", str(trials), os.path.join(base, "src", n + ".wav"), os.path.join(base, "ref_enc", n + ".aad")],
                   check=True, stdout=subprocess.DEVNULL)


def ref_decode(n):
    subprocess.run([ref, "-d", os.path.join(base, "ref_enc", n + ".aad"), os.path.join(base, "ref_dec", n + ".wav")],
                   check=True, stdout=subprocess.DEVNULL)


sample = names[: max(1, min(files, 4))]          # one core, a few files: scaled to the whole set
t = timed(lambda: [ref_encode(n) for n in sample])
out["reference_encode_one_core_s"] = round(t / len(sample) * files, 2)
t = timed(lambda: [ref_decode(n) for n in sample])
out["reference_decode_one_core_s"] = round(t / len(sample) * files, 2)
with ThreadPoolExecutor(16) as pool:
    out["reference_encode_16_processes_s"] = round(timed(lambda: list(pool.map(ref_encode, names))), 2)
    out["reference_decode_16_processes_s"] = round(timed(lambda: list(pool.map(ref_decode, names))), 2)
out["identical_aad"] = sha_dir(os.path.join(base, "gpu_enc")) == sha_dir(os.path.join(base, "ref_enc"))
out["identical_wav"] = sha_dir(os.path.join(base, "gpu_dec")) == sha_dir(os.path.join(base, "ref_dec"))
out["encode_speedup_vs_one_core"] = round(out["reference_encode_one_core_s"] / out["batch_encode_s"], 1)
out["encode_speedup_vs_16_processes"] = round(out["reference_encode_16_processes_s"] / out["batch_encode_s"], 1)
out["decode_speedup_vs_one_core"] = round(out["reference_decode_one_core_s"] / out["batch_decode_s"], 1)
out["decode_speedup_vs_16_processes"] = round(out["reference_decode_16_processes_s"] / out["batch_decode_s"], 1)
shutil.rmtree(base, ignore_errors=True)
print(json.dumps(out))
