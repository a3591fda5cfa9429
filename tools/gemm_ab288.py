"""A/B of the 288- and 256-tile GEMM kernels on shapes both tile exactly (N = 2304 = 8 x 288 = 9 x 256), bf16 epilogue.
Run twice: ESMK_GEMM_288=1 (default) and ESMK_GEMM_288=0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine

def main():
    lib = L.lib_experiments(); st = torch.cuda.current_stream().cuda_stream
    for M, N, K in ((32768, 2304, 1152), (32768, 2304, 3072), (32768, 2304, 9216), (32768, 4608, 1152)):
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
        print(f"288={os.environ.get('ESMK_GEMM_288','1')} M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
main()
