#!/bin/bash
# Measurement aid: the dense kernels on chip-filling batches of every (bits, channels) fast-path geometry
# (524288 recurrences = 8 dense waves per SIMD each), one JSON line per geometry.
for g in "4 2 262144" "3 2 262144" "2 2 262144" "4 1 524288" "3 1 524288" "2 1 524288"; do
  set -- $g
  echo "bits=$1 channels=$2 $(python tools/saturated_probe.py --bits $1 --channels $2 --streams $3 2>/dev/null | tail -1)"
done
