#!/bin/bash
# Measurement aid: kernel durations (rocprofv3 --kernel-trace --stats) of tools/host_api_bench for the
# in-tree library and a second build on the SAME box.  usage: tools/host_ab_profile.sh build/base [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hostab
rm -rf $O && mkdir -p $O
other=$1; shift
cd $R
for tag in new base; do
  if [ $tag = base ]; then export LD_LIBRARY_PATH=$R/$other; else unset LD_LIBRARY_PATH; fi
  rocprofv3 --kernel-trace --stats -f csv -d $O/$tag -- build/host_api_bench "$@" > $O/$tag.log 2>&1
  echo "=== $tag"; cut -c 1-700 $O/$tag.log | tail -2
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -d, -f1-7 "$f" | head -8
done
find $O -name "*.db" -delete
