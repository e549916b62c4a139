#!/bin/bash
# Measurement aid: the dense kernels on chip-filling batches of the fast-path geometries under two lane mappings
# (AAD_HIP_MAPPING) on the SAME box, two alternating rounds.  usage: tools/ab_mapping_saturated.sh <mapA> <mapB>
geos=${GEOS:-"4 2 262144;2 2 262144;4 1 524288;2 1 524288"}
IFS=';' read -ra G <<< "$geos"
for rep in 1 2; do
  for m in "$@"; do
    export AAD_HIP_MAPPING=$m
    for g in "${G[@]}"; do
      read -r bits ch streams <<< "$g"
      echo "$m bits=$bits ch=$ch $(python tools/saturated_probe.py --bits $bits --channels $ch --streams $streams --reps 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_ms %.4f dec_ms %.4f enc_gsps %.0f dec_gsps %.0f' % (d['encode_ms'], d['decode_ms'], d['encode_gsps'], d['decode_gsps']))")"
    done
  done
done
