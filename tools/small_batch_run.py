"""Runs one small-batch config a few times (for rocprofv3 --kernel-trace):  python tools/small_batch_run.py c1|c3b1|c3b2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ESMK_USE_X"):   # experiments library (ESMK_GEMM_* switches) in place of the product one
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from esm_amd import _lib as L
    L.lib = L.lib_experiments
import bench_configs as bc
print(bc.c1() if sys.argv[1] == "c1" else bc.c3(2 if sys.argv[1] == "c3b2" else 1, 512))
