#!/bin/bash
# Measurement aid: bench.py against library variants on the SAME box.  usage: tools/ab_bench.sh [variant.so ...]
# (default library first; each extra argument is a path handed to AAD_HIP_LIBRARY)
run() {
  python bench.py --no-saturated --no-cpu-baseline --no-extras $EXTRA 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['encode_msps'], d['decode_msps'], d['roofline']['kernel_ms'], d['roofline']['decode_kernel']['kernel_ms'], d['bit_exact_vs_reference_golden'])"
}
for rep in 1 2; do
  unset AAD_HIP_LIBRARY
  run default
  for v in "$@"; do
    export AAD_HIP_LIBRARY=$PWD/$v
    run $v
  done
done
