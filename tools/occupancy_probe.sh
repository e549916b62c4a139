#!/bin/bash
# Measurement aid (experiment): the saturated dense kernels with extra dynamic LDS per workgroup, i.e. fewer waves per SIMD
# (AAD_HIP_DEBUG_DYN_LDS, bytes).  One line per (geometry, bytes).
for g in "4 1 524288" "4 2 262144"; do
  set -- $g
  for dyn in 0 8000 13000 21000 34000 45000 61000; do
    echo "bits=$1 ch=$2 dyn=$dyn $(AAD_HIP_DEBUG_DYN_LDS=$dyn python tools/saturated_probe.py --bits $1 --channels $2 --streams $3 --reps 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_ms %.3f dec_ms %.3f enc_gsps %.0f dec_gsps %.0f' % (d['encode_ms'], d['decode_ms'], d['encode_gsps'], d['decode_gsps']))")"
  done
done
