#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   headline  = bench.py's step batch (1000 stereo one-block streams), saturated = tools/saturated_probe.py
#   (262 144 streams): kernel-trace statistics, SQ counters, and HBM traffic (FETCH_SIZE / WRITE_SIZE in
#   separate passes, never combined with a trace domain other than --kernel-trace).
# Outputs land under gpurun_out/prof/; tools/stamp_pmc.py folds them into profiles/${TAG}_pmc_stamp.json (the
# file bench.py quotes, stamped with the SHA-256 of aad_amd/csrc), tools/pmc_db_summary.py and
# tools/kernel_stats_db.py write the human-readable summaries.  usage: bash tools/collect_profiles.sh [tag]
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O/head $O/sat
cd $R
passes() {  # $1 = output dir, rest = the program
  local D=$1; shift
  rocprofv3 --kernel-trace --stats -d $D/kt -- "$@" > $D/kt.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $D/p1 -- "$@" > $D/p1.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $D/p2 -- "$@" > $D/p2.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $D/p3 -- "$@" > $D/p3.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $D/fetch -- "$@" > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $D/write -- "$@" > $D/write.log 2>&1
}
passes $O/head python3 bench.py --no-saturated --no-cpu-baseline --no-extras --no-config5
echo "headline passes done"
passes $O/sat python3 tools/saturated_probe.py
echo "saturated passes done"
python3 tools/kernel_stats_db.py $O/head/kt $O/${TAG}_bench_kernel_stats_rocprofv3.csv | head -6
python3 tools/kernel_stats_db.py $O/sat/kt $O/${TAG}_saturated_kernel_stats_rocprofv3.csv | head -6
python3 tools/pmc_db_summary.py $O/head 2 > $O/${TAG}_pmc_summary_bench.txt
python3 tools/pmc_db_summary.py $O/sat 1 > $O/${TAG}_saturated_pmc_summary.txt
python3 tools/stamp_pmc.py $O/${TAG}_pmc_stamp.json headline=$O/head:1000:992 saturated=$O/sat:262144:992
cp $O/${TAG}_pmc_stamp.json profiles/${TAG}_pmc_stamp.json   # so that the bench line below quotes THIS measurement
# the BASELINE shapes of the line's configs[] rows (cfg4's any-channel kernels, cfg2(ii), cfg5's shard), merged into the same stamp file
bash tools/collect_config_profiles.sh ${TAG} > $O/config_passes.log 2>&1 || { tail -20 $O/config_passes.log; exit 1; }
cp profiles/${TAG}_pmc_stamp.json $O/${TAG}_pmc_stamp.json
cd $R
# every kernel of the full line (trials 2 = the dual trial-search encoder, the other BASELINE shapes)
rocprofv3 --kernel-trace --stats -d $O/kt_full -- python3 bench.py --no-saturated --no-cpu-baseline > $O/kt_full.log 2>&1
python3 tools/kernel_stats_db.py $O/kt_full $O/${TAG}_bench_full_kernel_stats_rocprofv3.csv | head -14
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/bench_err.log
tail -c 800 $O/${TAG}_bench_line.json
# the raw rocpd databases are tens of MB per pass: only the summaries travel back
find $O -name "*.db" -delete
