"""Runs one small-batch config a few times (for rocprofv3 --kernel-trace):  python tools/small_batch_run.py c1|c3b1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
print(bc.c1() if sys.argv[1] == "c1" else bc.c3(1, 512))
