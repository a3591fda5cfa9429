"""Same-box comparison: esmk_gemm (bf16 epilogue) vs torch.matmul (hipBLASLt) on long-K and trunk shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
    shapes = [(16384, 8192, 8192), (32768, 2304, 9216), (32768, 4096, 4096), (32768, 3456, 1152), (32768, 1152, 3072), (32768, 6144, 1152)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for M, N, K in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        Wf = torch.randn(N, K) * 0.05
        Wp = engine.pack_weight(Wf, "cuda")
        Wb = Wf.cuda().bfloat16()
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        mine = t(lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_BF16, None, 1.0, out.data_ptr(), st)))
        vend = t(lambda: torch.matmul(A, Wb.t()))
        f = 2.0 * M * N * K / 1e9
        print(f"M={M} N={N} K={K}: esmk {mine*1e3:8.1f} us {f/mine:7.1f} TF | vendor {vend*1e3:8.1f} us {f/vend:7.1f} TF", flush=True)
main()
