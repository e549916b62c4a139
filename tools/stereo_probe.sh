#!/bin/bash
# Measurement aid: stereo geometries - GPU parity tests first, then the saturated probe (all three code widths) and the
# latency-bound 20 000-block decode for the built library and build/libaad_hip_base.so on the SAME box.
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/stereo_tests.log 2>&1 || { tail -30 gpurun_out/stereo_tests.log; exit 1; }
tail -1 gpurun_out/stereo_tests.log
for rep in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export AAD_HIP_LIBRARY=$PWD/build/libaad_hip_base.so; else unset AAD_HIP_LIBRARY; fi
    for bits in 4 3 2; do
      echo "$lib bits=$bits $(python tools/saturated_probe.py --bits $bits --channels 2 --streams 262144 --reps 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_ms %.3f dec_ms %.3f enc_gsps %.0f dec_gsps %.0f' % (d['encode_ms'], d['decode_ms'], d['encode_gsps'], d['decode_gsps']))")"
    done
    echo "$lib $(python tools/midsize_probe.py 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('cfg2ii_1000x16', 'cfg5_1250x10')})")"
  done
done
