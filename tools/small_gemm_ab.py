"""Small-batch GEMM shapes: product dispatch (split-K + reduce where it applies) against 128 x 64 tiles without split-K
(experiments library, ESMK_GEMM_BT64=1; one process per setting).   python tools/small_gemm_ab.py 0|1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ESMK_GEMM_BT64"] = sys.argv[1]
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib_experiments()
st = torch.cuda.current_stream().cuda_stream
SH = []
for M in (256, 512, 1024, 2048, 3072, 4096):
    for N, K in ((960, 960), (1152, 1152), (1536, 1536), (2560, 2560), (960, 2560), (1152, 3072), (1536, 4096), (2880, 960), (3456, 1152), (4608, 1536)):
        SH.append((f"M{M}", M, N, K, 0))
for name, M, N, K, epi in SH:
    A = torch.randn(M, K, device="cuda").bfloat16()
    Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda", L.PACK_SWIGLU if epi == 2 else L.PACK_PLAIN)
    out = torch.empty(M, N // 2 if epi == 2 else N, device="cuda", dtype=torch.bfloat16)
    wsb = lib.esmk_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    f = lambda: L.check(lib.esmk_gemm_ws(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 1.0, out.data_ptr(), ws.data_ptr(), wsb, st))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    if "flush" not in globals():
        flush = torch.empty(96 << 20, device="cuda")   # 384 MB written between launches: weights and activations come from HBM
    ts = []
    for _ in range(9):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"BT64={sys.argv[1]} {name:10s} {M}x{N}x{K}: {sorted(ts)[4]:6.1f} us  ({2.0 * M * N * K / sorted(ts)[4] / 1e6:5.0f} TF)  ws {wsb >> 20} MiB", flush=True)
