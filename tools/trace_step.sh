#!/bin/bash
# One kernel trace of the default step -> gpurun_out/trace_step/{trace.csv.gz,timeline.txt,alone.txt}; extra env via the environment.
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
OUT=gpurun_out/trace_step; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.log 2>&1 || exit 2
f=$(ls $OUT/raw/*/*kernel_trace.csv | head -1)
gzip -c "$f" > $OUT/trace.csv.gz; rm -rf $OUT/raw
python3 tools/timeline.py $OUT/trace.csv.gz > $OUT/timeline.txt 2>&1
python3 tools/alone_time.py $OUT/trace.csv.gz 25 > $OUT/alone.txt 2>&1
python3 tools/chain_gaps.py $OUT/trace.csv.gz 30 all > $OUT/gaps.txt 2>&1
