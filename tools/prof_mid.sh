export TMPDIR=/tmp
OUT=gpurun_out/mid
mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python3 tools/small_batch_run.py c3b2 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
cp $(find $OUT/t -name "*kernel_stats.csv" | head -n 1) $OUT/c3b2_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
tail -n 1 $OUT/run.log | cut -c1-200
