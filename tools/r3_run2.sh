#!/bin/bash
# round-3 GPU call 2: grouped weight gradients (test + sweeps), grid scaling by concurrency, CU mask inside 4 queues, parity tests, bench extras
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
export GPU_MAX_HW_QUEUES=8
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad_group or block_grads" > $OUT/pytest_r3b.log 2>&1; echo "ops pytest rc $?"; tail -3 $OUT/pytest_r3b.log
bash tools/gpu_sweep.sh "X=0" \
  "STLPOSE_WGRAD_GROUP=1" \
  "STLPOSE_WGRAD_GROUP=2" \
  "STLPOSE_WGRAD_GROUP=8" \
  "STLPOSE_WGRAD_GROUP=4 STLPOSE_WGRAD_BLOCKS=128" \
  "STLPOSE_WGRAD_GROUP=4 STLPOSE_WGRAD_BLOCKS=512" \
  "STLPOSE_WGRAD_GROUP=8 STLPOSE_WGRAD_BLOCKS=512" \
  "STLPOSE_CAP_SCALE=100,100,75,50" \
  "STLPOSE_CAP_SCALE=100,75,50,33" \
  "STLPOSE_CAP_SCALE=100,100,100,66" \
  "STLPOSE_CAP_SCALE=100,150,150,150" \
  "STLPOSE_STREAMS=3" \
  "STLPOSE_STREAMS=3 STLPOSE_WGRAD_STREAMS=n1" \
  "STLPOSE_STREAMS=3 STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:96" \
  "STLPOSE_STREAMS=3 STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:128" \
  "STLPOSE_SKIP_WGRAD=1" \
  "X=1" > $OUT/sweep2.txt 2>&1
echo sweep done; cat $OUT/sweep2.txt
timeout -k 10 900 python -m pytest tests/test_parity_r3_gpu.py -q > $OUT/pytest_r3a.log 2>&1; echo "parity pytest rc $?"
tail -5 $OUT/pytest_r3a.log
timeout -k 10 600 python bench.py --steps 30 --warmup 8 > $OUT/bench_r3b.json 2> $OUT/bench_r3b.err; echo "bench rc $?"
tail -c 300 $OUT/bench_r3b.err
