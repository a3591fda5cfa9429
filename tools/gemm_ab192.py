"""A/B of the 192-tile kernel vs the 256-tile path on the ESM3-open shapes at 8192 tokens (ESMK_GEMM_288=0 disables
the 288/192 kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine

def main():
    lib = L.lib_experiments(); st = torch.cuda.current_stream().cuda_stream
    for M, N, K in ((8192, 1536, 1536), (8192, 1536, 4096), (8192, 4608, 1536), (16384, 1536, 1536), (16384, 4608, 1536)):
        A = torch.randn(M, K, device="cuda").bfloat16()
        Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        run = lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_BF16, None, 1.0, out.data_ptr(), st))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"288/192={os.environ.get('ESMK_GEMM_288','1')} M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
main()
