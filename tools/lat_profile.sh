#!/bin/bash
# Measurement aid: LDS / VMEM / instruction-fetch latency counters of bench.py's headline kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/latprof
rm -rf $O && mkdir -p $O
cd $R
CMD="python3 bench.py --no-saturated --no-cpu-baseline --no-extras --steps 50 --warmup 5 $EXTRA"
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS -d $O/a -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES -d $O/b -- $CMD > $O/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH -d $O/c -- $CMD > $O/c.log 2>&1
python3 tools/pmc_db_summary.py $O 2
tail -3 $O/a.log
find $O -name "*.db" -delete
