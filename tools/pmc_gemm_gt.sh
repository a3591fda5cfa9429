# HBM-side read traffic of the dominant GEMM (FFN up-projection 32768 x 6144 x 1152, gemm256_kernel<SWIGLU>) against the
# panel width of its tile order (ESMK_GEMM_GT: the ~32 workgroups an XCD runs at once form a (32/gt channel tiles) x (gt token
# tiles) super-tile; product value 4), one rocprofv3 --pmc FETCH_SIZE pass per width over the experiments library, and the
# launch time of the same width without the profiler.   bash tools/pmc_gemm_gt.sh  -> gpurun_out/gemm_gt/r03_gemm_gt_traffic.txt
export TMPDIR=/tmp
export ESMK_LIB=libesmk_x.so
OUT=gpurun_out/gemm_gt
mkdir -p $OUT
R=$OUT/r03_gemm_gt_traffic.txt
echo "# gt  launch_us(no profiler)  FETCH_SIZE_KiB_raw  HBM-side read MB per launch (x2: gfx950 tallies 16-B/lane streaming reads at half size); algorithmic operand bytes 89.7 MB" > $R
for gt in 1 2 4 8 16 32; do
  export ESMK_GEMM_GT=$gt
  timeout -k 10 100 python3 tools/gemm_bench.py 30 ffn1_swiglu > $OUT/t_$gt.log 2>&1 || { echo "timing gt=$gt failed"; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/p_$gt -o p --output-format csv -- python3 tools/gemm_bench.py 10 ffn1_swiglu > $OUT/p_$gt.log 2>&1 || { echo "pmc gt=$gt failed"; exit 1; }
  us=$(grep ffn1_swiglu $OUT/t_$gt.log | awk '{for(i=1;i<=NF;i++) if($i=="us") print $(i-1)}')
  f=$(python3 tools/pmc_kernel.py gemm256_kernel $(find $OUT/p_$gt -name "*counter_collection.csv") | python3 -c "import json,sys; print(json.load(sys.stdin)['FETCH_SIZE'])")
  python3 -c "print('$gt', '$us', round($f), round($f*1024*2/1e6,1))" >> $R
done
cat $R
