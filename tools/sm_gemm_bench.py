"""Small-batch GEMM kernels (gemm_sm.hip) against the split-K dispatch they replace: correctness vs torch fp32 and time
with cold operands (384 MB written between launches).   python tools/sm_gemm_bench.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib_experiments() if os.environ.get("ESMK_USE_X") else L.lib()
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(96 << 20, device="cuda")


def timeit(f, n=9):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[n // 2]


def run(M, d, h):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, d, device="cuda", generator=g) * 2 + 0.3
    gam = torch.randn(d, device="cuda", generator=g) * 0.1 + 1
    bet = torch.randn(d, device="cuda", generator=g) * 0.1
    ws_bytes = 64 << 20
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    # ---- LN-prologue GEMMs: QKV (bf16 out) and FFN1 (SwiGLU)
    for name, N, epi in (("qkv", 3 * d, L.EPI_BF16), ("ffn1", 2 * h, L.EPI_SWIGLU)):
        Wf = torch.randn(N, d, device="cuda", generator=g) * 0.03
        mode = L.PACK_SWIGLU if epi == L.EPI_SWIGLU else L.PACK_PLAIN
        Wp = engine.pack_weight(Wf, "cuda", mode)
        wfold, u, v = torch.empty_like(Wf), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        L.check(lib.esmk_fold_layernorm(Wf.data_ptr(), gam.data_ptr(), bet.data_ptr(), N, d, wfold.data_ptr(), u.data_ptr(), v.data_ptr(), st))
        Wfp = engine.pack_weight(wfold, "cuda", mode)
        stats = torch.empty(M, 1, 2, device="cuda")
        xb = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
        L.check(lib.esmk_row_stats(x.data_ptr(), M, d, stats.data_ptr(), xb.data_ptr(), st))
        ow = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.empty(M, ow, device="cuda", dtype=torch.bfloat16)
        f_new = lambda: L.check(lib.esmk_gemm_lnfold(xb.data_ptr(), d, stats.data_ptr(), 1, 1e-5, Wfp.data_ptr(), u.data_ptr(),
                                                     v.data_ptr(), M, N, d, epi, out.data_ptr(), st))
        f_new(); torch.cuda.synchronize()
        xn = torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5)
        y = xn.bfloat16().float() @ Wf.bfloat16().float().T
        ref = torch.nn.functional.silu(y[:, :h]) * y[:, h:] if epi == L.EPI_SWIGLU else y
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        xnb = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
        out2 = torch.empty_like(out)
        wsb = lib.esmk_gemm_workspace_bytes(M, N, d)

        def f_old():
            L.check(lib.esmk_layernorm(x.data_ptr(), 0, gam.data_ptr(), bet.data_ptr(), xnb.data_ptr(), 1, M, d, 1e-5, st))
            L.check(lib.esmk_gemm_ws(xnb.data_ptr(), d, Wp.data_ptr(), M, N, d, epi, None, 1.0, out2.data_ptr(), ws.data_ptr(), wsb, st))
        t_new, t_old = timeit(f_new), timeit(f_old)
        print(f"M={M} d={d} {name:5s} N={N}: lnfold {t_new:6.1f} us   LN + gemm_ws {t_old:6.1f} us   rel err {err:.2e}", flush=True)
    # ---- residual GEMMs: out-proj (K = d) and FFN2 (K = h)
    for name, K in (("out", d), ("ffn2", h)):
        A = (torch.randn(M, K, device="cuda", generator=g)).bfloat16()
        Wf = torch.randn(d, K, device="cuda", generator=g) * 0.03
        Wp = engine.pack_weight(Wf, "cuda")
        parts = lib.esmk_res_stats_parts(d)
        stats = torch.zeros(M, parts, 2, device="cuda")
        xo = torch.empty_like(x)
        xb = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
        f_new = lambda: L.check(lib.esmk_gemm_residual_stats(A.data_ptr(), K, Wp.data_ptr(), M, d, K, None, 0.75, x.data_ptr(),
                                                             xo.data_ptr(), xb.data_ptr(), stats.data_ptr(), st))
        f_new(); torch.cuda.synchronize()
        ref = x + 0.75 * (A.float() @ Wf.bfloat16().float().T)
        err = (xo - ref).abs().max().item() / ref.abs().max().item()
        ssum = stats.sum(1)
        serr = max((ssum[:, 0] - xo.sum(1)).abs().max().item() / xo.sum(1).abs().max().item(),
                   (ssum[:, 1] - (xo * xo).sum(1)).abs().max().item() / (xo * xo).sum(1).abs().max().item())
        delta = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
        xn = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
        x2 = x.clone()
        wsb = lib.esmk_gemm_workspace_bytes(M, d, K)

        def f_old():
            L.check(lib.esmk_gemm_ws(A.data_ptr(), K, Wp.data_ptr(), M, d, K, L.EPI_BF16, None, 0.75, delta.data_ptr(), ws.data_ptr(), wsb, st))
            L.check(lib.esmk_add_layernorm(x2.data_ptr(), delta.data_ptr(), gam.data_ptr(), bet.data_ptr(), xn.data_ptr(), 1, M, d, 1e-5, st))
        t_new, t_old = timeit(f_new), timeit(f_old)
        print(f"M={M} d={d} {name:5s} K={K}: gemm_res {t_new:6.1f} us   gemm_ws + add_LN {t_old:6.1f} us   rel err {err:.2e} stats {serr:.1e}", flush=True)


for M in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    for d, h in ((1536, 4096), (960, 2560), (1152, 3072)):
        run(M, d, h)
