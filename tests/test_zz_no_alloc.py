"""esmk_gemm_noalloc (the entry point for callers that cannot allocate: split-K partials in the library's fixed device
array) against esmk_gemm_ws with a caller-owned workspace: same kernels, so the results must be bit-identical.  Two
calls in a row share the array, the second with a smaller problem; then a shape that is not split at all."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, L, engine, M, N, K, epi, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = torch.randn(N, K, generator=g) * 0.05
    swiglu = epi == L.EPI_SWIGLU
    Wp = engine.pack_weight(W, "cuda", L.PACK_SWIGLU if swiglu else L.PACK_PLAIN)
    bias = torch.randn(N, generator=g).cuda()
    s = torch.cuda.current_stream().cuda_stream
    shape = (M, N // 2) if swiglu else (M, N)
    dt = torch.float32 if epi == L.EPI_F32 else torch.bfloat16
    need = lib.esmk_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    ref = torch.full(shape, float("nan"), dtype=dt, device="cuda")
    L.check(lib.esmk_gemm_ws(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), 0.5, ref.data_ptr(),
                             ws.data_ptr(), need, s), "esmk_gemm_ws")
    out = torch.full(shape, float("nan"), dtype=dt, device="cuda")
    call = lambda: L.check(lib.esmk_gemm_noalloc(A.data_ptr(), K, Wp.data_ptr(), M, N, K, epi, bias.data_ptr(), 0.5,
                                                 out.data_ptr(), s), "esmk_gemm_noalloc")
    call()  # the documented one warm-up call per device (symbol lookup, LDS attribute)
    torch.cuda.synchronize()
    out.fill_(float("nan"))
    torch.cuda.synchronize()
    # driver-level free bytes: torch.cuda.memory_allocated() only sees torch's caching allocator and would miss a
    # hipMalloc / hipFree inside the library
    free_before = torch.cuda.mem_get_info()[0]
    call()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
    # and against plain fp32 arithmetic, so that "identical" cannot mean "identically wrong"
    y = A.float() @ W.bfloat16().float().cuda().T + bias
    want = torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if swiglu else (y if epi == L.EPI_F32 else 0.5 * y)
    if epi == L.EPI_F32:
        want = y
    assert (out.float() - want).abs().max().item() <= 0.01 * want.abs().max().item() + 2e-3
    return need


def test_gemm_noalloc_matches_workspace_form():
    from esm_amd import _lib as L
    from esm_amd import engine
    lib = L.lib()
    # split 8 ways (36 tiles, 48 K-steps): 18.9 MB of partials in the library's array
    assert _run(lib, L, engine, 512, 1152, 3072, L.EPI_BF16, 1) > 0
    # second call, smaller problem, other epilogue: reads nothing the first call left behind
    assert _run(lib, L, engine, 200, 384, 1024, L.EPI_F32, 2) > 0
    assert _run(lib, L, engine, 130, 512, 1152, L.EPI_SWIGLU, 3) > 0
    # not split (too few K-steps): no scratch involved, still the same result as the workspace form
    assert _run(lib, L, engine, 300, 256, 128, L.EPI_BF16, 4) == 0


def test_gemm_noalloc_rejects_bad_arguments():
    from esm_amd import _lib as L
    lib = L.lib()
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    assert lib.esmk_gemm_noalloc(x.data_ptr(), 64, x.data_ptr(), -1, 64, 64, L.EPI_BF16, None, 1.0, x.data_ptr(), None) == -1
    assert lib.esmk_gemm_noalloc(None, 64, x.data_ptr(), 64, 64, 64, L.EPI_BF16, None, 1.0, x.data_ptr(), None) == -1
    assert b"null pointer" in lib.esmk_last_error()
