set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sat
rm -rf $O
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
# row g1 (coalescing): requests between L1 (TCP), L2 (TCC) and the memory side, per launch
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/p7 -- python3 tools/saturated_probe.py > $O/p7.log 2>&1 || true
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/p8 -- python3 tools/saturated_probe.py > $O/p8.log 2>&1 || true
rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_REQ_sum TCC_NORMAL_WRITEBACK_sum -d $O/p9 -- python3 tools/saturated_probe.py > $O/p9.log 2>&1 || true
python3 tools/pmc_db_summary.py $O 1 > $R/gpurun_out/r02_saturated_pmc_summary.txt
cat $O/plain.json
# the raw rocpd databases are tens of MB per pass: only the summary travels back
find $O -name "*.db" -delete
