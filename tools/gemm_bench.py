"""Micro-benchmark of esmk_gemm on the trunk's shapes (random bf16 data).  Usage:
   python tools/gemm_bench.py [iters]            # TF/s table
Used under rocprofv3 --pmc for counter passes (keep iters small there)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
if os.environ.get("ESMK_LIB"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), os.environ["ESMK_LIB"])
from esm_amd import engine

SHAPES = [("ffn1_swiglu", 32768, 6144, 1152, L.EPI_SWIGLU), ("qkv_bf16", 32768, 3456, 1152, L.EPI_BF16),
          ("ffn2_resid", 32768, 1152, 3072, L.EPI_RESIDUAL), ("out_resid", 32768, 1152, 1152, L.EPI_RESIDUAL),
          ("ffn2_bf16", 32768, 1152, 3072, L.EPI_BF16), ("out_bf16", 32768, 1152, 1152, L.EPI_BF16),
          ("m8k_out", 8192, 1536, 1536, L.EPI_BF16), ("m8k_ffn2", 8192, 1536, 4096, L.EPI_BF16),
          ("m8k_qkv", 8192, 4608, 1536, L.EPI_BF16), ("m2k_qkv", 2048, 2880, 960, L.EPI_BF16),
          ("m2k_out", 2048, 960, 960, L.EPI_BF16), ("m2k_ffn1", 2048, 5120, 960, L.EPI_SWIGLU),
          ("m512_qkv", 512, 4608, 1536, L.EPI_BF16), ("m512_out", 512, 1536, 1536, L.EPI_BF16)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else None
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for name, M, N, K, epi in SHAPES:
        if only and only != name:
            continue
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = torch.randn(N, K) * 0.05
        Wp = engine.pack_weight(W, "cuda", L.PACK_SWIGLU if epi == L.EPI_SWIGLU else L.PACK_PLAIN)
        nout = N // 2 if epi == L.EPI_SWIGLU else N
        if only is None and not name.startswith(("ffn1", "qkv")) and "ALL" not in os.environ:
            pass
        out = torch.zeros(M, nout, device="cuda", dtype=torch.float32 if epi == L.EPI_RESIDUAL else torch.bfloat16)
        run = lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 0.01, out.data_ptr(), st))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:12s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
