import sys, os
sys.path.insert(0, "/root/repo")
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
for name, M, N, epi in (("ffn1 N=6144 swiglu", 32768, 6144, L.EPI_SWIGLU), ("N=3456 bf16 (288)", 32768, 3456, L.EPI_BF16), ("N=1152 bf16 (288)", 32768, 1152, L.EPI_BF16)):
    for K in (576, 1152, 2304, 4608):
        A = torch.randn(M, K, device="cuda").bfloat16()
        Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda", L.PACK_SWIGLU if epi == L.EPI_SWIGLU else L.PACK_PLAIN)
        out = torch.zeros(M, N // 2 if epi == L.EPI_SWIGLU else N, device="cuda", dtype=torch.bfloat16)
        run = lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 0.5, out.data_ptr(), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:20s} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
