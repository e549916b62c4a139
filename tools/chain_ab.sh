#!/bin/bash
# Measurement aid: the per-block chain of few long streams (tools/chain_probe.py) for the built library and
# build/libaad_hip_base.so on the SAME box.
for rep in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export AAD_HIP_LIBRARY=$PWD/build/libaad_hip_base.so; else unset AAD_HIP_LIBRARY; fi
    for args in "1 300 0" "1 300 2" "64 100 2" "512 20 2"; do
      echo "$lib $(python tools/chain_probe.py $args 2>/dev/null | tail -1)"
    done
  done
done
