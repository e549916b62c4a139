set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sat
mkdir -p $O
cd $R
python3 tools/saturated_probe.py > $O/plain.json
rocprofv3 --kernel-trace --stats -d $O/kt -- python3 tools/saturated_probe.py > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p1 -- python3 tools/saturated_probe.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/p2 -- python3 tools/saturated_probe.py > $O/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD -d $O/p3 -- python3 tools/saturated_probe.py > $O/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/p4 -- python3 tools/saturated_probe.py > $O/p4.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/p5 -- python3 tools/saturated_probe.py > $O/p5.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $O/p6 -- python3 tools/saturated_probe.py > $O/p6.log 2>&1 || true
cat $O/plain.json
find $O -name "*.csv" | head -30
