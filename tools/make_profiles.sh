#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates everything under profiles/ for round $1 (default r04).
# rocprofv3 needs TMPDIR=/tmp and the program itself after "--"; --pmc passes are separate runs with --kernel-trace only.
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-extras"
# 1. the bench line itself (default flags, with extras and CPU baselines)
timeout -k 10 700 python $ROOT/bench.py --dump-ops $OUT/${R}_ops_isolated.txt > $OUT/${R}_bench.json 2> $OUT/${R}_bench.stderr || exit 1
echo bench done
# 2. kernel traces: default multi-stream run and one stream (per-kernel durations not stretched by overlap)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- $B --steps 5 --warmup 3 > $OUT/trace_default.log 2>&1 || exit 2
STLPOSE_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- $B --steps 5 --warmup 3 > $OUT/trace_serial.log 2>&1 || exit 3
echo traces done
# 3. HBM counters, separate passes
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1 || exit 5
echo traffic passes done
# 4. MFMA utilisation counters, own pass
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $B --steps 2 --warmup 1 > $OUT/pmc_mfma.log 2>&1 || echo "mfma pass failed (rc $?)"
echo mfma pass done
# 5. one-rank RCCL rehearsal of the bucketed all-reduce: when does each bucket's collective start? (tools/dp_overlap.py)
STLPOSE_DP_FORCE=1 STLPOSE_BF16_BUCKETS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_dp -- $B --steps 5 --warmup 3 > $OUT/trace_dp.log 2>&1 || echo "dp trace failed (rc $?)"
# summaries (small; the raw traces stay on the box except gzipped kernel traces)
for d in trace_default trace_serial; do
  f=$(ls $OUT/$d/*/*kernel_trace.csv | head -1); gzip -c $f > $OUT/${R}_$d.csv.gz
  python $ROOT/tools/kernel_names.py $OUT/$d/*/*kernel_stats.csv > $OUT/${R}_${d}_kernel_stats.csv
  python $ROOT/tools/timeline.py $OUT/${R}_$d.csv.gz > $OUT/${R}_${d}_summary.txt
done
f=$(ls $OUT/trace_dp/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python $ROOT/tools/dp_overlap.py $f $OUT/${R}_dp_overlap.txt > /dev/null
python $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${R}_bench.json $OUT/${R}_pmc_dominant.json > /dev/null
python $ROOT/tools/step_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${R}_step_traffic.txt
SERIAL_MS=$(grep -o "sum of kernel time [0-9.]*" $OUT/${R}_trace_serial_summary.txt | grep -o "[0-9.]*$")
python $ROOT/tools/mfma_util.py $OUT/pmc_mfma $OUT/${R}_mfma_util.json $SERIAL_MS > $OUT/${R}_mfma_util.txt
python $ROOT/tools/alone_time.py $OUT/${R}_trace_default.csv.gz > $OUT/${R}_alone_time.txt 2>&1 || true
rm -rf $OUT/trace_default $OUT/trace_serial $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/trace_dp
ls -la $OUT
echo profiles done
# 6. (optional: `make_profiles.sh r04 sq`) SQ wave-state counters per kernel family: parked / issue-stalled / issuing share of the wave cycles
if [ "$2" = "sq" ]; then
  timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1 || echo "sq pass 1 failed"
  timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B --steps 2 --warmup 1 > $OUT/pmc_sq2.log 2>&1 || echo "sq pass 2 failed"
  python $ROOT/tools/pmc_sq.py $OUT/pmc_sq $OUT/pmc_sq2 > $OUT/${R}_sq_wave_states.txt
  rm -rf $OUT/pmc_sq $OUT/pmc_sq2
  echo sq done
fi
