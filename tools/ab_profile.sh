#!/bin/bash
# Measurement aid: rocprofv3 kernel durations + SQ counters of bench.py's headline kernels for library
# variants on the SAME box.  usage: tools/ab_profile.sh [variant.so ...]   (default library first)
# PMC passes carry no trace domain besides --kernel-trace (the pool refuses other combinations).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/abprof
rm -rf $O && mkdir -p $O
cd $R
CMD="python3 bench.py --no-saturated --no-cpu-baseline --no-extras --no-config5 --steps 50 --warmup 5 $EXTRA"
one() {
  tag=$1
  rocprofv3 --kernel-trace -d $O/$tag/kt -- $CMD > $O/$tag.kt.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/$tag/p1 -- $CMD > $O/$tag.p1.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/$tag/p2 -- $CMD > $O/$tag.p2.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/$tag/p3 -- $CMD > $O/$tag.p3.log 2>&1
  echo "=== $tag"
  python3 tools/pmc_db_summary.py $O/$tag 2
}
unset AAD_HIP_LIBRARY
one default
i=0
for v in "$@"; do
  i=$((i+1))
  export AAD_HIP_LIBRARY=$R/$v
  one variant$i
done
find $O -name "*.db" -delete
