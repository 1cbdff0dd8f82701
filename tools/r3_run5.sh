#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bnadd" > $OUT/pytest_r3c.log 2>&1; echo "bnadd pytest rc $?"; tail -4 $OUT/pytest_r3c.log
timeout -k 10 900 python -m pytest tests/test_hrnet_gpu.py tests/test_fullsize_gpu.py tests/test_parity_r2_gpu.py -x -q > $OUT/pytest_r3d.log 2>&1; echo "net pytest rc $?"; tail -6 $OUT/pytest_r3d.log
bash tools/gpu_sweep.sh "X=0" "STLPOSE_MERGE_BLOCK_END=0" "X=1" "STLPOSE_MERGE_BLOCK_END=0" 2>&1 | tee $OUT/sweep5.txt
