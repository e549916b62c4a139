#!/bin/bash
# Measurement aid: the dense kernels on chip-filling batches of every fast-path geometry for the built library and
# for library variants on the SAME box, two alternating rounds.  usage: tools/ab_saturated.sh [variant.so ...]
geos=${GEOS:-"4 2 262144;3 2 262144;2 2 262144;4 1 524288;3 1 524288;2 1 524288"}
libs=(default "$@")
IFS=';' read -ra G <<< "$geos"
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    if [ $lib = default ]; then unset AAD_HIP_LIBRARY; else export AAD_HIP_LIBRARY=$PWD/$lib; fi
    for g in "${G[@]}"; do
      read -r bits ch streams <<< "$g"
      echo "$lib bits=$bits ch=$ch $(python tools/saturated_probe.py --bits $bits --channels $ch --streams $streams --reps 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_ms %.4f dec_ms %.4f enc_gsps %.0f dec_gsps %.0f' % (d['encode_ms'], d['decode_ms'], d['encode_gsps'], d['decode_gsps']))")"
    done
  done
done
