#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 python -m pytest tests/test_trainer_gpu.py -x -q > $OUT/pytest_r3e.log 2>&1; echo "trainer pytest rc $?"; tail -15 $OUT/pytest_r3e.log
STLPOSE_DP_FORCE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_dpforce.json 2> $OUT/bench_dpforce.err; echo "dp-force rc $?"; tail -c 900 $OUT/bench_dpforce.json; tail -3 $OUT/bench_dpforce.err
STLPOSE_DP_FORCE=1 STLPOSE_BF16_BUCKETS=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_dpforce_bf16.json 2> $OUT/bench_dpforce_bf16.err; echo "dp-force bf16 rc $?"; tail -c 700 $OUT/bench_dpforce_bf16.json
STL_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc $?"; tail -c 900 $OUT/bench_gloo2.json; tail -3 $OUT/bench_gloo2.err
