#!/bin/bash
# SQ wave-state counters of one (serialised) step: where do wave cycles go -- parked (s_waitcnt / barrier), issue-stalled, or issuing?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
B="python $ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { echo "pmc2 failed"; tail -5 $OUT/pmc_sq2.log; }
python $ROOT/tools/pmc_sq.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/pmc_sq.txt; cat $OUT/pmc_sq.txt
rm -rf $OUT/pmc_sq $OUT/pmc_sq2
