#!/bin/bash
# PMC summary of the dense kernels on a chip-filling batch of EVERY fast-path geometry (bits 4/3/2 x stereo/mono):
# kernel-trace durations, VALU / LDS / VMEM instruction counts, VALU-active cycles, LDS bank conflicts, and HBM
# traffic (FETCH_SIZE and WRITE_SIZE in separate passes) against each side's algorithmic bytes.
# usage (through gpurun): bash tools/saturated_geometries_pmc.sh <tag>   -> gpurun_out/<tag>_saturated_geometries_pmc.txt
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/satgeo
rm -rf $O && mkdir -p $O
cd $R
IFS=';' read -ra GEO <<< "${GEOS:-4 2 262144;3 2 262144;2 2 262144;4 1 524288;3 1 524288;2 1 524288}"
for g in "${GEO[@]}"; do
  read -r bits ch streams <<< "$g"
  D=$O/b${bits}c${ch}
  mkdir -p $D
  A="--bits $bits --channels $ch --streams $streams --reps 3"
  python3 tools/saturated_probe.py $A > $D/plain.json
  rocprofv3 --kernel-trace -d $D/kt -- python3 tools/saturated_probe.py $A > $D/kt.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $D/p1 -- python3 tools/saturated_probe.py $A > $D/p1.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $D/p2 -- python3 tools/saturated_probe.py $A > $D/p2.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $D/p3 -- python3 tools/saturated_probe.py $A > $D/p3.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $D/p4 -- python3 tools/saturated_probe.py $A > $D/p4.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $D/p5 -- python3 tools/saturated_probe.py $A > $D/p5.log 2>&1
  echo "geometry bits=$bits channels=$ch done"
done
python3 tools/geometry_pmc_summary.py $O > $R/gpurun_out/${TAG}_saturated_geometries_pmc.txt
cat $R/gpurun_out/${TAG}_saturated_geometries_pmc.txt
find $O -name "*.db" -delete
