#!/bin/bash
# Measurement aid: bench.py's trials2 object (pipelined step with the reference CLI's default trial search) and
# tools/trial_probe.py for the built library and for build/libaad_hip_base.so, alternating on the SAME box.
run() {
  python bench.py --no-saturated --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['trials2']; print('$1', 'value', d['value'], 'trials2', t['value'], t['ms_per_step'], t['encode_ms'], t['decode_ms'], t['bit_exact_vs_reference_golden'])"
  python tools/trial_probe.py 2>/dev/null | head -4 | tr '\n' ' '; echo
}
for rep in 1 2; do
  unset AAD_HIP_LIBRARY
  run new
  export AAD_HIP_LIBRARY=$PWD/build/libaad_hip_base.so
  run base
done
