"""Structure encoder timing on one GPU (released geometry: d 1024, 128 vector heads, 2 blocks, 4096 codes; random
unit-scale weights).  One JSON line per (B, L).  Usage: python tools/vqenc_bench.py [--cases 1x512,16x512]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import synth  # noqa: E402
from esm_amd.vqvae import StructureTokenEncoder  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="1x512,16x512,4x2048")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    m = StructureTokenEncoder(1024, 1, 128, 2, 128, 4096)
    m.load_state_dict(synth.vqenc_weights(1024, 128, 2, 128, 4096, seed=3), strict=True)
    m = m.cuda()
    for case in args.cases.split(","):
        B, Ln = (int(v) for v in case.split("x"))
        coords = synth.synth_backbone(B, Ln, seed=B + Ln, nan_frac=0.05, jitter=0.15).cuda()
        ms = timed(lambda: m.encode(coords), n=args.iters)
        # MFMA work: per neighbourhood token and block 2 (d 15H + 3H d + d 2h + h d) flops
        d, H, h = 1024, 128, 4096
        flops = B * Ln * 16 * 2 * 2 * (d * 15 * H + 3 * H * d + d * 2 * h + h * d)
        print(json.dumps({"B": B, "L": Ln, "encode_ms": round(ms, 3), "residues_per_s": round(B * Ln / ms * 1e3),
                          "gemm_tflops": round(flops / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
