#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
export GPU_MAX_HW_QUEUES=8
bash tools/gpu_sweep.sh "X=0" "STLPOSE_WGRAD_GROUP=1 STLPOSE_WGRAD_BLOCKS=256" "STLPOSE_WGRAD_BLOCKS=768" "STLPOSE_WGRAD_GROUP=2" "STLPOSE_WGRAD_GROUP=3" "STLPOSE_WGRAD_GROUP=4 STLPOSE_WGRAD_BLOCKS_K1=256" "STL_WGRAD_64=0" "X=1" > $OUT/sweep3.txt 2>&1
cat $OUT/sweep3.txt
timeout -k 10 900 python -m pytest tests/test_parity_r3_gpu.py -q > $OUT/pytest_r3a.log 2>&1; echo "parity pytest rc $?"
tail -5 $OUT/pytest_r3a.log
bash tools/make_profiles.sh r03 > $OUT/make_profiles_r03.log 2>&1; echo "profiles rc $?"; tail -3 $OUT/make_profiles_r03.log
