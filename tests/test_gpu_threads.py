"""Threading contract of the drop-in boundary: the reference keeps no mutable globals (static const tables,
reference src/aad_tables.c:8,58), so DISTINCT handles are usable from DISTINCT threads.  The legacy layer here borrows its HIP
contexts from a process-wide pool behind a mutex (aad_amd/csrc/aad_legacy_api.c, eight slots): eight threads x fifty rounds of the
reference CLI's two call sequences (reference src/main.c:182-198, :91-106), each on handles of its own, mixed parameters, every
result byte-equal to the oracle - from Python threads (ctypes releases the GIL inside a call) and from a C99 pthreads program
(tests/c/threads_legacy.c).  More threads than pool slots in a third run, so that contexts are created and destroyed, not only
parked."""
import os
import shutil
import subprocess
import threading

import numpy as np
import pytest

import oracle_binding as ob
from aad_amd.synth import synth_pcm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _jobs(thread, rounds):
    rng = np.random.default_rng(4200 + thread)
    jobs = []
    for k in range(rounds):
        ch = int(rng.integers(1, 3))
        bits = int(rng.choice([4, 3, 2]))
        ms = bool(ch == 2 and rng.integers(0, 4) == 0)
        trials = int(rng.choice([0, 0, 1, 2]))
        mbs = int(rng.choice([1024, 256, 600, 2048]))
        n = int(rng.integers(1, 6000))
        pcm = synth_pcm(1, n, ch, seed=900 + thread, first_stream=k, kind=str(rng.choice(["music", "noise"])))[0]
        image = ob.encode(pcm, bits, mbs, 48000, ms, trials)
        jobs.append((pcm, bits, mbs, ms, trials, image, ob.decode(image)[0]))
    return jobs


@pytest.mark.parametrize("threads,rounds", [(8, 50), (12, 12)])
def test_distinct_handles_from_distinct_threads(threads, rounds):
    import aad_amd
    codec = aad_amd.LegacyCodec(aad_amd.load_library())
    work = [_jobs(t, rounds) for t in range(threads)]  # expected results first: the threads then spend their time in the library
    errors, start = [], threading.Barrier(threads)

    def run(t):
        try:
            start.wait()
            for k, (pcm, bits, mbs, ms, trials, image, decoded) in enumerate(work[t]):
                got = codec.encode(pcm, bits, mbs, 48000, ms, trials)  # Create -> SetEncodeParameter -> EncodeWhole -> Destroy
                if got != image:
                    errors.append((t, k, "encode", bits, mbs, ms, trials, pcm.shape))
                    continue
                out, _ = codec.decode(got)                             # Create -> DecodeHeader -> DecodeWhole -> Destroy
                if not np.array_equal(out, decoded):
                    errors.append((t, k, "decode", bits, mbs, ms, trials, pcm.shape))
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    pool = [threading.Thread(target=run, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    assert not errors, errors[:5]


def test_c_pthreads_on_distinct_handles(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("needs gcc")
    lib_dir, oracle_dir = os.path.join(ROOT, "aad_amd"), os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "threads_legacy")
    build = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + oracle_dir,
                            "-o", exe, os.path.join(ROOT, "tests", "c", "threads_legacy.c"), "-L" + lib_dir, "-laad_hip", "-L" + oracle_dir,
                            "-laad_oracle", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + oracle_dir, "-Wl,-rpath,/opt/rocm/lib"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr[-2000:])
    assert run.stdout.startswith("ok: 8 threads x 50 rounds"), run.stdout
