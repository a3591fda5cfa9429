# Counters of the attention kernel (plain instantiation, back to back): two rocprofv3 --pmc passes of <= 8 SQ counters
# (never mixed with tracing) over tools/attn_bench.py at the C2 shape and at C5's L = 4096, averaged per launch by
# tools/pmc_kernel.py.   bash tools/pmc_attn.sh   -> gpurun_out/pmc_attn/r03_pmc_attn_{c2,c5}.json
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
for shape in c2 c5; do
  it=8; [ $shape = c5 ] && it=4
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/${shape}_p1 -o p --output-format csv -- python3 tools/attn_bench.py $shape $it > $OUT/${shape}_p1.log 2>&1 || { echo "pass 1 $shape failed"; tail -3 $OUT/${shape}_p1.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM -d $OUT/${shape}_p2 -o p --output-format csv -- python3 tools/attn_bench.py $shape $it > $OUT/${shape}_p2.log 2>&1 || { echo "pass 2 $shape failed"; tail -3 $OUT/${shape}_p2.log; exit 1; }
  python3 tools/pmc_kernel.py attn_varlen $(find $OUT/${shape}_p1 $OUT/${shape}_p2 -name "*counter_collection.csv") > $OUT/r03_pmc_attn_$shape.json
  cat $OUT/r03_pmc_attn_$shape.json
done
