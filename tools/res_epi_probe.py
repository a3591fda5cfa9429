import os, sys
sys.path.insert(0, "/root/repo")
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib_experiments(); st = torch.cuda.current_stream().cuda_stream
M, N = 32768, 1152
for K in (1152, 3072):
    A = torch.randn(M, K, device="cuda").bfloat16()
    Wp = engine.pack_weight(torch.randn(N, K) * 0.05, "cuda")
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    of = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    for name, epi, out in (("bf16", L.EPI_BF16, ob), ("residual", L.EPI_RESIDUAL, of), ("f32", L.EPI_F32, of)):
        run = lambda: L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, None, 1.0, out.data_ptr(), st))
        for _ in range(3): run()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"288={os.environ.get('ESMK_GEMM_288','1')} K={K} {name}: {best*1e3:.1f} us", flush=True)
