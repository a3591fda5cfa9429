import os, sys
sys.path.insert(0, "tools")
import bench_configs as bc
for L in (64, 128, 256, 512):
    r = bc.c3(1, L)
    print(L, round(r["ms_per_forward"], 3), flush=True)
