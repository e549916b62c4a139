"""INTEGRATION.md section 2's two-context pipeline as a C program (tests/c/pipeline_signal.c): plain C99 against include/ and
libaad_hip.so - AADHip_ContextSignalNextRun, device-resident plans on two streams, hipStreamWaitEvent between them - built here
with gcc and run on the GPU; it checks every step's images and PCM against the host-memory entry points itself."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_pipeline_with_attached_events(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    lib_dir = os.path.join(ROOT, "aad_amd")
    exe = str(tmp_path / "pipeline_signal")
    build = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                            "-o", exe, os.path.join(ROOT, "tests", "c", "pipeline_signal.c"), "-L" + lib_dir, "-laad_hip", "-L/opt/rocm/lib",
                            "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert run.stdout.startswith("ok:"), run.stdout
