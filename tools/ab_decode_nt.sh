#!/bin/bash
# Same-box A/B: the dense stereo 4-bit decoder with streamed (non-temporal) PCM stores against plain stores, by batch size, and the
# write traffic of both on BASELINE config 2(ii)'s shape.  usage (gpurun): bash tools/ab_decode_nt.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/ab_decode_nt.txt
: > $O
for spec in "1000 16" "1250 10" "8000 1" "20000 1" "40000 2" "65536 1" "100000 1" "150000 1"; do
  set -- $spec
  for nt in 0 1000000000; do
    line=$(AAD_HIP_DECODE_NT_MIN=$nt python3 tools/saturated_probe.py --streams $1 --blocks $2 --reps 20 2>/dev/null | tail -1)
    echo "streams=$1 blocks=$2 nt_min=$nt $line" | cut -c1-230 | tee -a $O
  done
done
for nt in 0 1000000000; do
  rm -rf gpurun_out/nt_w$nt
  AAD_HIP_DECODE_NT_MIN=$nt rocprofv3 --pmc WRITE_SIZE -d gpurun_out/nt_w$nt -- python3 tools/saturated_probe.py --streams 1000 --blocks 16 > /dev/null 2>&1
  python3 - <<PY | tee -a $O
import sys
sys.path.insert(0, "tools")
from stamp_pmc import collect
k = collect("gpurun_out/nt_w$nt")
for name, e in k.items():
    if "decode" in name:
        print("nt_min=$nt", name.split("(")[0], "WRITE_SIZE bytes per launch", int(e.get("WRITE_SIZE", 0) * 1024), "= %.3f x PCM bytes" % (e.get("WRITE_SIZE", 0) * 1024 / 63488000.0))
PY
  find gpurun_out/nt_w$nt -name "*.db" -delete
done
