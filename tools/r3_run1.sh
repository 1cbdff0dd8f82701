#!/bin/bash
# round-3 GPU call 1: CU-mask placement probe, calibration + CU-mask sweeps, new parity tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
export GPU_MAX_HW_QUEUES=8
timeout -k 10 120 python tools/cumask_probe.py > $OUT/cumask_probe.txt 2>&1; echo "probe rc $?"
bash tools/gpu_sweep.sh "X=0" \
  "STLPOSE_SKIP_WGRAD=1" \
  "STLPOSE_STREAMS=1 STLPOSE_WGRAD_STREAMS=0" \
  "STLPOSE_WGRAD_STREAMS=n1" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:64" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:96" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:128" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:128 STLPOSE_WGRAD_BLOCKS=128" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:64 STLPOSE_WGRAD_BLOCKS=128" \
  "STLPOSE_WGRAD_STREAMS=n2 STLPOSE_CUMASK_OFF=0:128" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:64 STLPOSE_CUMASK_CHAIN=64:256" \
  "STLPOSE_WGRAD_STREAMS=n1 STLPOSE_CUMASK_OFF=0:128 STLPOSE_CUMASK_CHAIN=128:256 STLPOSE_WGRAD_BLOCKS=128" \
  "STLPOSE_CUMASK_CHAIN=0:256" \
  "X=1" > $OUT/sweep1.txt 2>&1
echo sweep done
timeout -k 10 900 python -m pytest tests/test_parity_r3_gpu.py -x -q > $OUT/pytest_r3a.log 2>&1; echo "pytest rc $?"
tail -5 $OUT/pytest_r3a.log
timeout -k 10 600 python bench.py --steps 30 --warmup 8 > $OUT/bench_r3a.json 2> $OUT/bench_r3a.err; echo "bench rc $?"
tail -c 600 $OUT/bench_r3a.err
