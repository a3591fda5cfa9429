"""A/B of attention-kernel builds in ONE process (interleaved rounds): the product library and variant builds
(make VARIANT=name ... -> esm_amd/libesmk_x_name.so).   python tools/attn_ab.py name [name ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from esm_amd import _lib as L
from esm_amd.packing import pack_lengths

libs = [("product", L.lib())] + [(n, L._load(os.path.join(os.path.dirname(L.LIB_PATH), f"libesmk_x_{n}.so"))) for n in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
for name, lens, H in (("c2 64x512 H18", [512] * 64, 18), ("c5 4x4096 H40", [4096] * 4, 40),
                      ("c4 varlen H18", list(np.random.default_rng(0).integers(64, 2048, 40)), 18), ("c1 8x256 H15", [256] * 8, 15)):
    d = 64 * H
    pb = pack_lengths(lens)
    N = pb.n_tokens
    qkv = (torch.randn(N, 3 * d, device="cuda") * 0.6).bfloat16()   # |q.k| * log2e/8-scale scores like the model's
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
    cu = dev(pb.cu_seqlens)
    cu_h = np.ascontiguousarray(pb.cu_seqlens, dtype=np.int32)
    maps = {}
    for n, lib in libs:   # every library plans with its own query-block size
        nq = lib.esmk_attn_plan(cu_h.ctypes.data, len(cu_h) - 1, None, 0)
        m = np.zeros((nq, 2), dtype=np.int32)
        lib.esmk_attn_plan(cu_h.ctypes.data, len(cu_h) - 1, m.ctypes.data, nq)
        maps[n] = (dev(m.reshape(-1)), nq)
    outs = [torch.empty(N, d, device="cuda", dtype=torch.bfloat16) for _ in libs]
    names = {id(lib): n for n, lib in libs}
    run = lambda lib, o: L.check(lib.esmk_attn_varlen(qkv.data_ptr(), o.data_ptr(), cu.data_ptr(), maps[names[id(lib)]][0].data_ptr(),
                                                      maps[names[id(lib)]][1], N, d, H, st))
    for (n, lib), o in zip(libs, outs):
        for _ in range(3):
            run(lib, o)
    torch.cuda.synchronize()
    times = [[] for _ in libs]
    for r in range(7):
        for i, (n, lib) in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(lib, outs[i])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 20 * 1e3)
    flops = sum(4.0 * l * l * d for l in lens)
    line = f"{name:14s}"
    for (n, lib), t, o in zip(libs, times, outs):
        med = sorted(t)[len(t) // 2]
        diff = (o.float() - outs[0].float()).abs().max().item()
        line += f" | {n} {med:7.1f} us {flops / med / 1e6:6.0f} TF maxdiff {diff:.3g}"
    print(line, flush=True)
