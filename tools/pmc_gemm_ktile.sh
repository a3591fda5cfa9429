# Stall attribution of the dominant GEMM (gemm256_kernel<SWIGLU>, FFN up-projection 32768 x 6144 x 1152) from hardware
# counters: four rocprofv3 --pmc passes (never mixed with tracing; <= 8 SQ / 4 TCP / 2 TA / 2 GRBM counters per pass) over
# the kernel micro-benchmark, then tools/pmc_gemm_ktile_parse.py -> gpurun_out/gemm_ktile/r03_pmc_gemm_ktile.json.
export TMPDIR=/tmp
OUT=gpurun_out/gemm_ktile
mkdir -p $OUT
CMD="tools/gemm_bench.py 12 ffn1_swiglu"
pass() {  # name counters...
  n=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $OUT/$n -o p --output-format csv -- python3 $CMD > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $OUT/$n.log; exit 1; }
}
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
pass p2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TA_TA_BUSY TA_BUFFER_READ_LDS_WAVEFRONTS &&
pass p3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES &&
pass p4 SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM &&
python3 tools/pmc_gemm_ktile_parse.py $OUT > $OUT/r03_pmc_gemm_ktile.json && tail -n 60 $OUT/r03_pmc_gemm_ktile.json
