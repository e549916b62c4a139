"""The HIP encoder replaces the quantiser's integer division by convert / fma / truncate
(aad_amd/csrc/aad_encode.hip.h encode_step).  tests/quantiser_equiv.c proves the two equal for
every reachable operand; the GPU side of the same claim is covered by the parity tests."""
import os
import subprocess


def test_fma_quantiser_equals_integer_division(tmp_path):
    src = os.path.join(os.path.dirname(__file__), "quantiser_equiv.c")
    exe = tmp_path / "qe"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == 3 * 256 * (98304 + 17) and int(out[1]) == 0
