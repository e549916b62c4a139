#!/bin/bash
# Measurement aid: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and VALU occupancy of the dense kernels on a
# chip-filling batch of one geometry.  usage (through gpurun): bash tools/saturated_traffic.sh <bits> <channels> <streams> <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sat_$4
rm -rf $O && mkdir -p $O
cd $R
A="--bits $1 --channels $2 --streams $3 --reps 3"
python3 tools/saturated_probe.py $A > $O/plain.json
rocprofv3 --pmc FETCH_SIZE -d $O/p4 -- python3 tools/saturated_probe.py $A > $O/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p5 -- python3 tools/saturated_probe.py $A > $O/p5.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/p2 -- python3 tools/saturated_probe.py $A > $O/p2.log 2>&1
python3 tools/pmc_db_summary.py $O 1 > $O/summary.txt
cat $O/plain.json $O/summary.txt
find $O -name "*.db" -delete
