#!/bin/bash
# Collect the round's rocprofv3 evidence for bench.py's workload on the GPU box (run through gpurun):
#   kernel-trace statistics, SQ counters, and HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
#   passes, never combined with a trace domain).  Outputs land under gpurun_out/prof/ and are
#   summarised into profiles/ by tools/kernel_stats_db.py, tools/pmc_db_summary.py, tools/hbm_traffic.py.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd $R
CMD="python3 bench.py --no-saturated --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/kt -- $CMD > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/p3 -- $CMD > $O/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- $CMD > $O/write.log 2>&1
python3 tools/kernel_stats_db.py $O/kt $O/kernel_stats.csv | head -6
# every kernel of the full line (trials 2 = the dual trial-search encoder, the other BASELINE shapes)
rocprofv3 --kernel-trace --stats -d $O/kt_full -- python3 bench.py --no-saturated --no-cpu-baseline > $O/kt_full.log 2>&1
python3 tools/kernel_stats_db.py $O/kt_full $O/kernel_stats_full.csv | head -12
python3 tools/pmc_db_summary.py $O 2 > $O/pmc_summary.txt
python3 tools/hbm_traffic.py $O/fetch $O/write 1000 992 $O/hbm_traffic.json
cp $O/hbm_traffic.json profiles/r02_hbm_traffic.json   # so that the bench line below quotes THIS measurement
python3 bench.py > $O/bench_line.json 2> $O/bench_err.log
tail -c 600 $O/bench_line.json
# the raw rocpd databases are tens of MB per pass: only the summaries travel back
find $O -name "*.db" -delete
