#!/bin/bash
# usage: gpu_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench line (ms/step) per configuration
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  out=$(env $cfg timeout -k 10 200 python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "$cfg -> $out"
done
