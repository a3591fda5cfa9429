"""Micro-benchmark of esmk_qk_prep + esmk_attn_varlen on the C2 / C5 shapes (random data)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from esm_amd import _lib as L
if os.environ.get("ESMK_LIB"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["ESMK_LIB"])
from esm_amd import engine
from esm_amd.packing import pack_lengths

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
ONLY = sys.argv[1] if len(sys.argv) > 1 else None   # "c2" / "c5" / "varlen": one shape only (PMC passes)
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for name, lens, H in (("c2 64x512 H18", [512] * 64, 18), ("c5 4x4096 H40", [4096] * 4, 40), ("varlen H18", list(np.random.default_rng(0).integers(64, 2048, 40)), 18)):
    if ONLY and not name.startswith(ONLY):
        continue
    d = 64 * H
    pb = pack_lengths(lens)
    N = pb.n_tokens
    qkv = torch.randn(N, 3 * d, device="cuda").bfloat16()
    ctx = torch.empty(N, d, device="cuda", dtype=torch.bfloat16)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
    cu, qmap, pos = dev(pb.cu_seqlens), dev(pb.qblk_map.reshape(-1)), dev(pb.positions)
    g = torch.ones(d, device="cuda")
    cos, sin = engine.rotary_tables(int(pb.positions.max()) + 1)
    cos, sin = cos.cuda(), sin.cuda()
    def prep():
        L.check(lib.esmk_qk_prep(qkv.data_ptr(), g.data_ptr(), g.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), cos.shape[0], N, d, H, 0.18, 1e-5, st))
    def attn():
        L.check(lib.esmk_attn_varlen(qkv.data_ptr(), ctx.data_ptr(), cu.data_ptr(), qmap.data_ptr(), pb.qblk_map.shape[0], N, d, H, st))
    flops = sum(4.0 * l * l * d for l in lens)
    for fn, nm, fl in ((prep, "qk_prep", 0), (attn, "attention", flops)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / ITERS
        extra = f"{fl/ms/1e9:8.1f} TF/s" if fl else f"{N*2*d*2*2/ms/1e6:8.1f} GB/s"
        print(f"{name:16s} {nm:10s} {ms*1e3:8.1f} us {extra}", flush=True)
