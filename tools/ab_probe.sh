#!/bin/bash
# Measurement aid: GPU tests, then tools/midsize_probe.py and tools/saturated_probe.py for the built library and for
# build/libaad_hip_base.so (a build of an earlier commit) alternating on the SAME box.
set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2_gpu_tests6.log 2>&1 || { tail -30 gpurun_out/s2_gpu_tests6.log; exit 1; }
tail -2 gpurun_out/s2_gpu_tests6.log
for rep in 1 2; do
  unset AAD_HIP_LIBRARY
  echo "new  $(timeout -k 10 120 python tools/midsize_probe.py 2>/dev/null | tail -1)"
  echo "new  sat $(timeout -k 10 120 python tools/saturated_probe.py 2>/dev/null | tail -1)"
  export AAD_HIP_LIBRARY=$PWD/build/libaad_hip_base.so
  echo "base $(timeout -k 10 120 python tools/midsize_probe.py 2>/dev/null | tail -1)"
  echo "base sat $(timeout -k 10 120 python tools/saturated_probe.py 2>/dev/null | tail -1)"
done
