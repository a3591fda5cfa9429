"""Mid-size GEMM shapes around the 256-tile / 128-tile crossover, operands from HBM: default dispatch against a forced
tile family (experiments library, ESMK_GEMM_TILE=128|256; one process per setting).  python tools/mid_gemm_ab.py 0|128|256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ESMK_GEMM_TILE"] = sys.argv[1]
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib_experiments()
st = torch.cuda.current_stream().cuda_stream
SH = []
for M in (1024, 2048, 3072, 4096, 6144, 8192):
    for N, K, epi in ((5120, 960, 2), (6144, 1152, 2), (8192, 1536, 2), (2880, 960, 0), (3456, 1152, 0), (4608, 1536, 0)):
        SH.append((M, N, K, epi))
flush = torch.empty(96 << 20, device="cuda")
for M, N, K, epi in SH:
    A = torch.randn(M, K, device="cuda").bfloat16()
    Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda", L.PACK_SWIGLU if epi == 2 else L.PACK_PLAIN)
    out = torch.empty(M, N // 2 if epi == 2 else N, device="cuda", dtype=torch.bfloat16)
    wsb = lib.esmk_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    f = lambda: L.check(lib.esmk_gemm_ws(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 1.0, out.data_ptr(), ws.data_ptr(), wsb, st))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"TILE={sys.argv[1]} {M}x{N}x{K} epi{epi}: {sorted(ts)[4]:6.1f} us", flush=True)
