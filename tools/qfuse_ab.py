"""QKV projection -> attention context, two ways in ONE process (interleaved rounds):
  A  esmk_gemm + esmk_qk_prep + esmk_attn_varlen                       (three passes over q)
  B  esmk_gemm_qkv_stats + esmk_k_prep + esmk_attn_varlen_qfused       (q normalised / rotated inside the attention kernel)
Per-call times by events too.   python tools/qfuse_ab.py [rounds]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from esm_amd import _lib as L
from esm_amd import engine
from esm_amd.packing import pack_lengths

lib = L.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
st = torch.cuda.current_stream().cuda_stream
q_scale = 0.125 * math.log2(math.e)
for name, lens, H in (("c2 64x512 d1152", [512] * 64, 18), ("c3 16x512 d1536", [512] * 16, 24),
                      ("c4 varlen d1152", list(np.random.default_rng(0).integers(64, 2048, 31)), 18),
                      ("300M 64x512 d960", [512] * 64, 15), ("300M 16x512 d960", [512] * 16, 15),
                      ("6B 32x512 d2560", [512] * 32, 40), ("c5 4x4096 d2560", [4096] * 4, 40)):
    d = 64 * H
    pb = pack_lengths(lens)
    M = pb.n_tokens
    parts = lib.esmk_qkv_stats_parts(M, d)
    if not parts:
        print(name, "no fused path for this shape")
        continue
    xn = torch.randn(M, d, device="cuda").bfloat16()
    Wp = engine.pack_weight(torch.randn(3 * d, d) * 0.04, "cuda")
    gq, gk = torch.ones(d, device="cuda"), torch.ones(d, device="cuda")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
    pos, cu, qmap = dev(pb.positions), dev(pb.cu_seqlens), dev(pb.qblk_map.reshape(-1))
    cos, sin = [t.cuda() for t in engine.rotary_tables(int(pb.positions.max()) + 1)]
    nq = pb.qblk_map.shape[0]
    qkv = torch.empty(M, 3 * d, device="cuda", dtype=torch.bfloat16)
    ctx = [torch.empty(M, d, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    stats = torch.empty(M, parts, 2, device="cuda")
    qmr = torch.empty(M, 2, device="cuda")
    A = [lambda: L.check(lib.esmk_gemm(xn.data_ptr(), d, Wp.data_ptr(), M, 3 * d, d, L.EPI_BF16, None, 1.0, qkv.data_ptr(), st)),
         lambda: L.check(lib.esmk_qk_prep(qkv.data_ptr(), gq.data_ptr(), gk.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                          cos.shape[0], M, d, H, q_scale, 1e-5, st)),
         lambda: L.check(lib.esmk_attn_varlen(qkv.data_ptr(), ctx[0].data_ptr(), cu.data_ptr(), qmap.data_ptr(), nq, M, d, H, st))]
    B = [lambda: L.check(lib.esmk_gemm_qkv_stats(xn.data_ptr(), d, Wp.data_ptr(), M, d, qkv.data_ptr(), stats.data_ptr(), st)),
         lambda: L.check(lib.esmk_k_prep(qkv.data_ptr(), gk.data_ptr(), None, pos.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                         cos.shape[0], M, d, H, 1e-5, stats.data_ptr(), parts, qmr.data_ptr(), st)),
         lambda: L.check(lib.esmk_attn_varlen_qfused(qkv.data_ptr(), ctx[1].data_ptr(), cu.data_ptr(), qmap.data_ptr(), nq, M, d, H, 0,
                                                     qmr.data_ptr(), gq.data_ptr(), None, pos.data_ptr(), cos.data_ptr(),
                                                     sin.data_ptr(), cos.shape[0], q_scale, st))]
    for chain in (A, B):
        for _ in range(3):
            for f in chain:
                f()
    torch.cuda.synchronize()
    diff = (ctx[0].float() - ctx[1].float()).abs().max().item()
    tot = {0: [], 1: []}
    per = {0: [[], [], []], 1: [[], [], []]}
    for r in range(rounds):
        for i, chain in enumerate((A, B)):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            acc = [0.0, 0.0, 0.0]
            for _ in range(10):
                ev[0].record()
                for j, f in enumerate(chain):
                    f()
                    ev[j + 1].record()
                torch.cuda.synchronize()
                for j in range(3):
                    acc[j] += ev[j].elapsed_time(ev[j + 1]) * 1e3 / 10
            for j in range(3):
                per[i][j].append(acc[j])
            tot[i].append(sum(acc))
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{name}: M={M} parts={parts} maxdiff {diff:.3g}")
    for i, lab in enumerate(("A three-call", "B q-fused  ")):
        print(f"   {lab}: gemm {med(per[i][0]):6.1f}  prep {med(per[i][1]):6.1f}  attn {med(per[i][2]):6.1f}  total {med(tot[i]):6.1f} us", flush=True)
