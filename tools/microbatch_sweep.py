"""Does splitting the 64 x 512 step into micro-batches (smaller working set -> Infinity Cache hits for the
HBM-bound passes) pay?  Prints ms per 32768 residues for several micro-batch sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import synth
from esm_amd.engine import DeviceBatch
from esm_amd.packing import pack_lengths
from tools.bench_configs import esmc, timed

m = esmc(1152, 18, 36)
for mb in (64, 32, 16, 8, 4):
    tok = synth.synth_tokens([512] * mb).reshape(-1).cuda()
    batch = DeviceBatch(pack_lengths([512] * mb), "cuda")
    n = 64 // mb
    def fn():
        for _ in range(n):
            m.forward_packed(tok, batch)
    dt = timed(fn, 5, 2)
    print(f"micro-batch {mb:3d} x 512 ({n} per step): {dt*1e3:7.2f} ms/step  {32768/dt:10.0f} residues/s", flush=True)
