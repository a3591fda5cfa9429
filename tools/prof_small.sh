# kernel-trace of the small-batch forwards (C3 1 x 512, C1 8 x 256):  bash tools/prof_small.sh [tag]
export TMPDIR=/tmp
TAG=${1:-base}
OUT=gpurun_out/small_$TAG
mkdir -p $OUT
for c in c3b1 c1; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/$c -o t --output-format csv -- python3 tools/small_batch_run.py $c > $OUT/$c.log 2>&1 || { echo "$c failed"; tail -5 $OUT/$c.log; exit 1; }
  cp $(find $OUT/$c -name "*kernel_stats.csv" | head -n 1) $OUT/${c}_kernel_stats.csv
  tail -n 1 $OUT/$c.log
  find $OUT/$c -name "*kernel_trace.csv" -size +20M -delete
done
