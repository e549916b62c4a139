#!/bin/bash
# Same-box A/B of the dense encoders' byte ring on MID-SIZE one-block batches (16 385 .. 65 536 lanes: one-wave workgroups), the
# regime round 3 left unmeasured: `static` = a library built from the previous commit (ring area for four waves in
# static LDS: 72 864 B mono / 54 432 B stereo 4-bit per one-wave workgroup), `dynamic` = this tree (one wave's worth of ring rows
# per wave, dynamic LDS), `noring` = this tree with AAD_HIP_ENCODE_RING=0.  usage (gpurun): bash tools/ab_ring_midsize.sh <old .so>
set -e
OLD=$(realpath ${1:-build/ab/libaad_hip_static_ring.so})
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_ring_midsize.txt
: > $O
run() { # label, env..., -- args
  local label=$1; shift
  local line
  line=$(env "$@" 2>/dev/null | tail -1)
  echo "$label $line" | tee -a $O
}
for spec in "1 20000" "1 40000" "1 48000" "1 64000" "2 20000" "2 28000" "2 32000"; do
  set -- $spec
  CH=$1; N=$2
  ARGS="python3 tools/saturated_probe.py --streams $N --channels $CH --bits 4 --reps 20"
  run "ch=$CH streams=$N static " AAD_HIP_LIBRARY=$OLD $ARGS
  run "ch=$CH streams=$N dynamic" $ARGS
  run "ch=$CH streams=$N noring " AAD_HIP_ENCODE_RING=0 $ARGS
done
