export TMPDIR=/tmp
OUT=gpurun_out/c1_small
mkdir -p $OUT
export ESMK_SMALL_TOKENS=2048
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python3 tools/small_batch_run.py c1 > $OUT/c1.log 2>&1 || { echo failed; tail -5 $OUT/c1.log; exit 1; }
cp $(find $OUT/t -name "*kernel_stats.csv" | head -n 1) $OUT/c1_small_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
tail -n 1 $OUT/c1.log
