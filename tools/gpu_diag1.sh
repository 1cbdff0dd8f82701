#!/bin/bash
# GPU box: baseline bench + kernel timelines (default multi-stream and serial) for offline gap analysis.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/diag1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || exit 1
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_default -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/trace_default.log 2>&1 || exit 2
STLPOSE_STREAMS=1 STLPOSE_WGRAD_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_serial -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/trace_serial.log 2>&1 || exit 3
for d in trace_default trace_serial; do f=$(ls $OUT/$d/*/*kernel_trace.csv | head -1); cut -d, -f1-20 $f | gzip > $OUT/$d.csv.gz; rm -rf $OUT/$d; done
ls -la $OUT
echo diag1 done
