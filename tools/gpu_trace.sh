#!/bin/bash
# GPU box: kernel timeline of the default bench (steps 3) -> gpurun_out/$1.csv.gz ; extra env passes through
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-trace}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/_$NAME -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/$NAME.log 2>&1 || exit 2
f=$(ls $OUT/_$NAME/*/*kernel_trace.csv | head -1); gzip -c $f > $OUT/$NAME.csv.gz; rm -rf $OUT/_$NAME
echo trace $NAME done
