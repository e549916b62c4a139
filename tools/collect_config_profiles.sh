#!/bin/bash
# rocprofv3 evidence for the BASELINE shapes on bench.py's `configs[]` rows (run through gpurun, after tools/collect_profiles.sh
# or on its own): cfg4 (10 000 eight-channel one-block segments, 3- and 2-bit: the any-channel kernels), cfg2(ii) (1000 stereo
# 4-bit streams x 16 blocks) and cfg5's per-GPU shard (1250 files x 10 blocks) - the same passes as the headline (kernel
# trace; SQ counters; FETCH_SIZE and WRITE_SIZE each in a pass of its own, never next to another trace domain).
# tools/stamp_pmc.py --merge folds them into profiles/${TAG}_pmc_stamp.json as workloads cfg4_3bit, cfg4_2bit, cfg2ii, cfg5_shard.
# usage: bash tools/collect_config_profiles.sh [tag] [quick]      quick = instruction counts and traffic only (3 passes)
set -e
TAG=${1:-r04}
QUICK=${2:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg
rm -rf $O && mkdir -p $O
cd $R
passes() {  # $1 = output dir, rest = the program
  local D=$1; shift
  mkdir -p $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $D/p1 -- "$@" > $D/p1.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $D/fetch -- "$@" > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $D/write -- "$@" > $D/write.log 2>&1
  if [ -z "$QUICK" ]; then
    rocprofv3 --kernel-trace --stats -d $D/kt -- "$@" > $D/kt.log 2>&1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $D/p2 -- "$@" > $D/p2.log 2>&1
    rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $D/p3 -- "$@" > $D/p3.log 2>&1
  fi
}
passes $O/cfg4_3bit python3 tools/saturated_probe.py --streams 10000 --channels 8 --bits 3
echo "cfg4 3-bit passes done"
passes $O/cfg4_2bit python3 tools/saturated_probe.py --streams 10000 --channels 8 --bits 2
echo "cfg4 2-bit passes done"
if [ -z "$QUICK" ]; then
  passes $O/cfg2ii python3 tools/saturated_probe.py --streams 1000 --blocks 16
  echo "cfg2(ii) passes done"
  passes $O/cfg5_shard python3 tools/saturated_probe.py --streams 1250 --blocks 10
  echo "cfg5 shard passes done"
  python3 tools/stamp_pmc.py --merge profiles/${TAG}_pmc_stamp.json cfg4_3bit=$O/cfg4_3bit:10000:292 cfg4_2bit=$O/cfg4_2bit:10000:444 \
      cfg2ii=$O/cfg2ii:1000:15872 cfg5_shard=$O/cfg5_shard:1250:9920 > $O/stamp.log
  cp profiles/${TAG}_pmc_stamp.json $O/${TAG}_pmc_stamp.json
else
  python3 tools/stamp_pmc.py $O/${TAG}_quick_stamp.json cfg4_3bit=$O/cfg4_3bit:10000:292 cfg4_2bit=$O/cfg4_2bit:10000:444 > $O/stamp.log
fi
cat $O/stamp.log
find $O -name "*.db" -delete
