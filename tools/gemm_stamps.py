"""Diagnostic: per-segment s_memtime stamps of gemm256 (needs esm_amd/libesmk_stamps.so built with -DESMK_STAMPS)."""
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
raw.esmk_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.esmk_debug_stamps(buf.ctypes.data, 8 * 64)
buf = buf.reshape(8, 64)[:, :7].astype(np.int64)
base = buf.min()
names = ["b1s", "b1e", "P", "b2e", "b3e", "Q", "b4e"]
print("wave " + " ".join(f"{n:>5s}" for n in names))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{int(v - base):5d}" for v in buf[w]))
