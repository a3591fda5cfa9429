# Profiles of the headline bench (run on the GPU box from the repo root):  bash tools/collect_profiles.sh
#   1. rocprofv3 --kernel-trace --stats over bench.py          -> gpurun_out/prof/bench_kernel_stats.csv
#   2. three separate --pmc passes over bench.py (never mixed with tracing; <= 4 TCC / 8 SQ / 2 GRBM counters per pass)
#   3. tools/pmc_bench_parse.py                                 -> gpurun_out/prof/r03_pmc_bench.json (carries esmk_build_id)
# Copy the two result files into profiles/ afterwards (bench.py reads profiles/r03_pmc_bench.json).
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
B="bench.py --steps 3 --warmup 2 --no-cpu-baseline"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 $B > $OUT/trace.log 2>&1 || { echo "trace pass failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 $B > $OUT/pmc_fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o p --output-format csv -- python3 $B > $OUT/pmc_write.log 2>&1 || { echo "WRITE_SIZE pass failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p --output-format csv -- python3 $B > $OUT/pmc_sq.log 2>&1 || { echo "SQ pass failed"; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -n 1) $OUT/bench_kernel_stats.csv
python3 tools/pmc_bench_parse.py $OUT > $OUT/r03_pmc_bench.json && tail -n 30 $OUT/r03_pmc_bench.json

# ---- the other BASELINE.json configs: one-line results + a kernel-trace summary each (bash tools/collect_profiles.sh all)
if [ "$1" = "all" ]; then
  timeout -k 10 900 python3 tools/bench_configs.py c1 c3 c3b1 c4 c5 2> $OUT/other.err | grep '^{' > $OUT/r03_other_configs.jsonl || { echo "bench_configs failed"; tail -3 $OUT/other.err; exit 1; }
  cat $OUT/r03_other_configs.jsonl
  timeout -k 10 300 python3 tools/generate_bench.py 2>/dev/null | grep '^{' > $OUT/r03_generate.jsonl; cat $OUT/r03_generate.jsonl
  for c in c1 c3b1; do
    timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace_$c -o t --output-format csv -- python3 tools/small_batch_run.py $c > $OUT/trace_$c.log 2>&1 || { echo "trace $c failed"; exit 1; }
    cp $(find $OUT/trace_$c -name "*kernel_stats.csv" | head -n 1) $OUT/r03_${c}_kernel_stats.csv
  done
  for c in c3 c4 c5; do
    timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_$c -o t --output-format csv -- python3 tools/bench_configs.py $c > $OUT/trace_$c.log 2>&1 || { echo "trace $c failed"; exit 1; }
    cp $(find $OUT/trace_$c -name "*kernel_stats.csv" | head -n 1) $OUT/r03_${c}_kernel_stats.csv
  done
  find $OUT -name "*kernel_trace.csv" -delete
fi
