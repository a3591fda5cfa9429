"""Diagnostic: per-tile stamps of the persistent GEMM (libesmk_stamps.so built with -DESMK_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from esm_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libesmk_stamps.so")
from esm_amd import engine
lib = L.lib()
M, N, K = 32768, 6144, 1152
A = torch.randn(M, K, device="cuda").bfloat16()
Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda", L.PACK_SWIGLU)
out = torch.zeros(M, N // 2, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_SWIGLU, None, 1.0, out.data_ptr(), st))
torch.cuda.synchronize()
buf = np.zeros(8 * 64, dtype=np.uint64)
raw = ctypes.CDLL(L.LIB_PATH)
raw.esmk_debug_pstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.esmk_debug_pstamps(buf.ctypes.data, 8 * 64)
buf = buf.reshape(8, 64).astype(np.int64)
for w in (0, 4):
    b = buf[w]
    base = b[0]
    print(f"wave {w}: per tile (start, kloop_end, epi_end) relative:")
    for j in range(11):
        s0, e0, e1 = b[3 * j] - base, b[3 * j + 1] - base, b[3 * j + 2] - base
        nxt = b[3 * (j + 1)] - base
        print(f"  tile {j}: start {s0:8d}  kloop {e0 - s0:7d}  epilogue {e1 - e0:6d}  next-start-gap {nxt - e1:6d}")
