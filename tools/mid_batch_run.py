"""Forwards of 1024-token batches (ESM3-open-shaped 2 x 512 with structure track; ESM C 300M-shaped 4 x 256; ESM C
600M-shaped 2 x 512): ms per forward.   python tools/mid_batch_run.py [n: batches of n x 512 tokens]   (ESMK_SMALL_TOKENS=n
moves the small-batch path's limit)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from esm_amd import synth
from esm_amd.engine import DeviceBatch
from esm_amd.packing import pack_lengths

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2   # batch = NB x 512 tokens
r = bc.c3(NB, 512)
print(f"esm3-open {NB}x512", round(r["ms_per_forward"], 3), flush=True)
for d, H, NL, B, Ln in ((960, 15, 30, 2 * NB, 256), (1152, 18, 36, NB, 512)):
    m = bc.esmc(d, H, NL)
    tok = synth.synth_tokens([Ln] * B).reshape(-1).cuda()
    batch = DeviceBatch(pack_lengths([Ln] * B), "cuda")
    dt = bc.timed(lambda: m.forward_packed(tok, batch), 20, 5)
    print(f"esmc d{d} {B}x{Ln}", round(dt * 1e3, 3), flush=True)
    del m
    torch.cuda.empty_cache()
