"""Which channel-tile width should esmk_gemm pick?  Times the bf16 epilogue on shapes whose width divides by both 288 and 192
(N % 576 == 0: the QKV projection of d_model 960 / 1152, the d 1152 down-projections) under the dispatch switches of the
process's environment; run it once per setting and compare:
    python tools/gemm_tile_pick.py                       # the product dispatch
    ESMK_GEMM_192=1 python tools/gemm_tile_pick.py       # 192-wide tiles wherever N % 192 == 0
    ESMK_GEMM_288=0 python tools/gemm_tile_pick.py       # 256-wide tiles only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine


def main():
    lib = L.lib_experiments(); st = torch.cuda.current_stream().cuda_stream
    tag = "192" if os.environ.get("ESMK_GEMM_192") == "1" else "256" if os.environ.get("ESMK_GEMM_288") == "0" else "dflt"
    shapes = [(M, 2880, 960) for M in (4096, 8192, 12288, 16384, 32768)] + \
             [(M, 3456, 1152) for M in (4096, 8192, 12288, 16384, 32768, 35078)] + \
             [(M, 1152, K) for M in (8192, 16384, 32768, 35078) for K in (1152, 3072)]
    for M, N, K in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        run = lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_BF16, None, 1.0, out.data_ptr(), st))
        for _ in range(3): run()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"{tag} M={M} N={N} K={K}: {best*1e3:8.1f} us {2.0*M*N*K/best/1e9:7.1f} TF/s", flush=True)


main()
