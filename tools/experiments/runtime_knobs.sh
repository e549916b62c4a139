#!/bin/bash
# Experiment: HIP runtime environment knobs against the headline (K = 20 as the driver runs it, and K = 200).  Every run has its own
# 60 s limit and the script stops at the first run that fails or times out; progress goes to gpurun_out/runtime_knobs.txt.
# usage (GPU box): bash tools/experiments/runtime_knobs.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/runtime_knobs.txt
: > $OUT
run() {
  local label=$1; shift
  for k in 20 200; do
    env "$@" timeout -k 5 60 python3 bench.py --steps $k --warmup 5 --no-saturated --no-cpu-baseline --no-config5 --no-extras 2>/dev/null > gpurun_out/knob_line.json || { echo "$label K=$k: failed or timed out - stopping" >> $OUT; return 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/knob_line.json').read().strip().splitlines()[-1])
print('%-36s K=%-3d %9.1f Msamples/s  %.5f ms/step  enc %.5f' % ('$label', $k, d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> $OUT
  done
}
# (ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT in the list any more: the two-stream pipeline never finishes under it - see README.md)
run "default" AAD_NOP=1 && run "ROC_ACTIVE_WAIT_TIMEOUT=100000" ROC_ACTIVE_WAIT_TIMEOUT=100000 && run "default (again)" AAD_NOP=1
cat $OUT
