#!/bin/bash
# round-3 closing run: full GPU suite, smoke, default bench (timed), profiles r03
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/pytest_r3_full.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_r3_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); python bench.py > $OUT/bench_r3_default.json 2> $OUT/bench_r3_default.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s"
STLPOSE_DP_FORCE=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline > $OUT/bench_dpforce.json 2>/dev/null; echo "dp-force rc $?"
bash tools/make_profiles.sh r03 > $OUT/make_profiles_r03.log 2>&1; echo "profiles rc $?"; tail -2 $OUT/make_profiles_r03.log
