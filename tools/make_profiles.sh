#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates everything under profiles/ for round $1 (default r01).
# rocprofv3 needs TMPDIR=/tmp and the program itself after "--".
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line itself (default flags)
timeout -k 10 400 python $ROOT/bench.py > $OUT/${R}_bench.json 2> $OUT/${R}_bench.stderr || exit 1
# 2. kernel trace, default multi-stream run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python $ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/trace_default.log 2>&1 || exit 2
# 3. kernel trace, one stream (per-kernel durations not stretched by overlap)
STLPOSE_STREAMS=1 STLPOSE_WGRAD_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python $ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/trace_serial.log 2>&1 || exit 3
# 4. HBM counters, separate passes
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 5
# keep the merged-back output small: stats + a gzip of the traces
for d in trace_default trace_serial; do cp $OUT/$d/*/*kernel_stats.csv $OUT/${R}_${d}_kernel_stats.csv; done
echo profiles done
