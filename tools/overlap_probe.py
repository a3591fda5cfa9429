"""Does an HBM-bound row kernel hide behind an MFMA-bound GEMM when both run at once on two streams?
Times (a) the FFN up-projection GEMM alone, (b) add+LayerNorm alone, (c) both concurrently on half batches.
   python tools/overlap_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine


def main():
    lib = L.lib()
    M, d, h2 = 16384, 1152, 6144
    A = torch.randn(M, d, device="cuda").bfloat16()
    Wp = engine.pack_weight(torch.randn(h2, d) * 0.05, "cuda", L.PACK_SWIGLU)
    out = torch.zeros(M, h2 // 2, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(M, d, device="cuda")
    delta = torch.randn(M, d, device="cuda").bfloat16()
    g = torch.ones(d, device="cuda")
    b = torch.zeros(d, device="cuda")
    y = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def gemm(st):
        L.check(lib.esmk_gemm(A.data_ptr(), d, Wp.data_ptr(), M, h2, d, L.EPI_SWIGLU, None, 1.0, out.data_ptr(), st.cuda_stream))

    def ln(st):
        L.check(lib.esmk_add_layernorm(x.data_ptr(), delta.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(),
                                       L.BF16, M, d, 1e-5, st.cuda_stream))

    def timed(fn, iters=40):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        s1.synchronize()
        s2.synchronize()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    import time

    def wall(fn, iters=40):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    tg = wall(lambda: gemm(s1))
    tl = wall(lambda: ln(s2))
    tl4 = wall(lambda: [ln(s2) for _ in range(2)])
    both = wall(lambda: (gemm(s1), ln(s2)))
    both2 = wall(lambda: (gemm(s1), ln(s2), ln(s2)))
    print(f"gemm alone {tg:.1f} us   add+LN alone {tl:.1f} us (x2: {tl4:.1f})   gemm || LN {both:.1f} us   gemm || 2xLN {both2:.1f} us")


if __name__ == "__main__":
    main()
