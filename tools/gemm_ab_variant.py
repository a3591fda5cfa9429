"""A/B of a variant build (make VARIANT=name -> esm_amd/libesmk_x_name.so) against the product library in ONE process:
interleaved rounds on the trunk's GEMM shapes, outputs compared bit for bit.
   python tools/gemm_ab_variant.py <variant> [rounds] [shape substring]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine

variant = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3] if len(sys.argv) > 3 else ""
base = L.lib()
var = L._load(os.path.join(os.path.dirname(L.LIB_PATH), f"libesmk_x_{variant}.so"))
print("base", base.esmk_build_id().decode(), "variant", variant, flush=True)
SHAPES = [("ffn1_swiglu", 32768, 6144, 1152, L.EPI_SWIGLU), ("qkv", 32768, 3456, 1152, L.EPI_BF16),
          ("ffn2", 32768, 1152, 3072, L.EPI_BF16), ("out", 32768, 1152, 1152, L.EPI_BF16),
          ("c3_ffn1", 8192, 8192, 1536, L.EPI_SWIGLU), ("c5_ffn1", 16384, 13824, 2560, L.EPI_SWIGLU),
          ("ragged", 5000, 6144, 1152, L.EPI_SWIGLU), ("k64", 8192, 1024, 64, L.EPI_BF16), ("k128", 8192, 4608, 128, L.EPI_SWIGLU)]
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K, epi in SHAPES:
    if only not in name:
        continue
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K) * 0.05
    Wp = engine.pack_weight(W, "cuda", L.PACK_SWIGLU if epi == L.EPI_SWIGLU else L.PACK_PLAIN)
    nout = N // 2 if epi == L.EPI_SWIGLU else N
    outs = [torch.zeros(M, nout, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    run = lambda lib, o: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 0.5, o.data_ptr(), st))
    for lib, o in ((base, outs[0]), (var, outs[1])):
        for _ in range(3):
            run(lib, o)
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    times = {0: [], 1: []}
    for r in range(rounds):
        for i, lib in enumerate((base, var)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(lib, outs[i])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 20 * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    b, v = med(times[0]), med(times[1])
    print(f"{name:12s} M={M} N={N} K={K}: base {b:7.1f} us (min {min(times[0]):7.1f})  {variant} {v:7.1f} us (min {min(times[1]):7.1f})  "
          f"{(b / v - 1) * 100:+5.1f} %  identical={same}", flush=True)
