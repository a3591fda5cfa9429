"""The small-batch path (gemm_sm.hip; esmk_trunk.small_tokens): LayerNorm-folded projections (esmk_gemm_lnfold), residual
epilogues that emit the next LayerNorm's statistics (esmk_gemm_residual_stats), 6 launches per block.

Kernel level: against plain torch fp32 (reference ops: attention.py:26-28,49; blocks.py:30-38,146-155).
Model level: the same model on the small-batch path and on the large-batch path (small_tokens = 0), both against the CPU
oracle - gate of the north star: per-residue cosine >= 0.999, logits within 0.05 + 0.02 max|logit|."""
import pytest
import torch

from esm_amd import _lib as L
from esm_amd import engine, synth

pytestmark = pytest.mark.gpu


def _st():
    return torch.cuda.current_stream().cuda_stream


def _fold(lib, W, gam, bet, mode):
    N, K = W.shape
    wf, u = torch.empty_like(W), torch.empty(N, device="cuda")
    v = torch.empty(N, device="cuda") if bet is not None else None
    L.check(lib.esmk_fold_layernorm(W.data_ptr(), gam.data_ptr(), L.ptr(bet), N, K, wf.data_ptr(), u.data_ptr(), L.ptr(v), _st()))
    return engine.pack_weight(wf, "cuda", mode), u, v


@pytest.mark.parametrize("M,d,N,epi,offset", [
    (512, 1536, 4608, "bf16", 0.3), (200, 960, 2880, "bf16", 0.0), (77, 1152, 3456, "f32", -0.5),
    (512, 1536, 8192, "swiglu", 0.2),   # 256 tiles of 128 x 128: the wide-tile instantiation
    (130, 960, 5120, "swiglu", 0.0), (300, 128, 360, "f32", 0.1),
    (256, 1152, 3456, "bf16", 25.0),    # |mean| = 12 sigma: the mean's share of the accumulator must cancel
    (1000, 1536, 4608, "bf16", 0.1), (1024, 960, 5120, "swiglu", -0.2),   # >= 280 tiles: two stages, two workgroups per CU
    (900, 1152, 3840, "f32", 0.0),
])
def test_lnfold_projection_vs_torch(M, d, N, epi, offset):
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, d, device="cuda", generator=g) * 2 + offset
    x[:, 7] *= 30  # an outlier channel
    gam = torch.randn(d, device="cuda", generator=g) * 0.2 + 1
    bet = torch.randn(d, device="cuda", generator=g) * 0.2 if epi != "f32" else None  # f32 = the geometric projection: no beta
    W = torch.randn(N, d, device="cuda", generator=g) * 0.03
    code = {"bf16": L.EPI_BF16, "f32": L.EPI_F32, "swiglu": L.EPI_SWIGLU}[epi]
    Wp, u, v = _fold(lib, W, gam, bet, L.PACK_SWIGLU if epi == "swiglu" else L.PACK_PLAIN)
    # statistics the way the trunk gets them: from a residual epilogue (x = x0 + 0 . W^T keeps x, emits parts) ...
    parts = lib.esmk_res_stats_parts(d)
    stats = torch.full((M, parts, 2), float("nan"), device="cuda")
    xb = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    if d % 64 == 0 and M != 77:
        A0 = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)
        W0 = engine.pack_weight(torch.zeros(d, 64), "cuda")
        xo = torch.empty_like(x)
        L.check(lib.esmk_gemm_residual_stats(A0.data_ptr(), 64, W0.data_ptr(), M, d, 64, None, 1.0, x.data_ptr(), xo.data_ptr(),
                                             xb.data_ptr(), stats.data_ptr(), _st()))
        assert torch.equal(xo, x)
    else:  # ... or from esmk_row_stats (the embedding output in front of block 0)
        parts = 1
        stats = torch.empty(M, 1, 2, device="cuda")
        L.check(lib.esmk_row_stats(x.data_ptr(), M, d, stats.data_ptr(), xb.data_ptr(), _st()))
    assert torch.equal(xb, x.bfloat16())
    ow = N // 2 if epi == "swiglu" else N
    out = torch.full((M, ow), float("nan"), device="cuda", dtype=torch.float32 if epi == "f32" else torch.bfloat16)
    L.check(lib.esmk_gemm_lnfold(xb.data_ptr(), d, stats.data_ptr(), parts, 1e-5, Wp.data_ptr(), u.data_ptr(), L.ptr(v), M, N, d,
                                 code, out.data_ptr(), _st()), "esmk_gemm_lnfold")
    torch.cuda.synchronize()
    y = torch.nn.functional.layer_norm(x.double(), (d,), gam.double(), None if bet is None else bet.double(), 1e-5) @ W.double().T
    ref = (torch.nn.functional.silu(y[:, :N // 2]) * y[:, N // 2:] if epi == "swiglu" else y).float()
    assert torch.isfinite(out.float()).all()
    err = (out.float() - ref).abs().max().item()
    # bf16 operands (x and W'), fp32 accumulation: the same class of error as bf16(LN(x)) . bf16(W)^T
    yb = torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5).bfloat16().double() @ W.bfloat16().double().T
    refb = (torch.nn.functional.silu(yb[:, :N // 2]) * yb[:, N // 2:] if epi == "swiglu" else yb).float()
    base = (refb - ref).abs().max().item()
    lim = (8.0 if offset > 5 else 3.0) * base + 2e-3 * ref.abs().max().item()
    assert err <= lim, (err, base, lim)


@pytest.mark.parametrize("M,N,K,inplace", [(512, 1536, 4096, True), (200, 960, 960, False), (65, 1152, 3072, False),
                                           (300, 128, 768, True)])
def test_residual_epilogue_with_stats_vs_torch(M, N, K, inplace):
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    Wp = engine.pack_weight(W, "cuda")
    x = torch.randn(M, N, device="cuda", generator=g) * 3 + 1
    bias = torch.randn(N, device="cuda", generator=g)
    ref = x.double() + 0.8 * (A.double() @ W.bfloat16().double().T + bias.double())
    parts = lib.esmk_res_stats_parts(N)
    stats = torch.full((M, parts, 2), float("nan"), device="cuda")
    xb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    xo = x if inplace else torch.full_like(x, float("nan"))
    keep = x.clone()
    L.check(lib.esmk_gemm_residual_stats(A.data_ptr(), K, Wp.data_ptr(), M, N, K, bias.data_ptr(), 0.8, x.data_ptr(), xo.data_ptr(),
                                         xb.data_ptr(), stats.data_ptr(), _st()), "esmk_gemm_residual_stats")
    torch.cuda.synchronize()
    if not inplace:
        assert torch.equal(x, keep)
    assert (xo.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert torch.equal(xb, xo.bfloat16())
    tot = stats.double().sum(1)
    assert (tot[:, 0] - xo.double().sum(1)).abs().max().item() <= 1e-4 * xo.abs().sum(1).max().item()
    assert (tot[:, 1] - (xo.double() ** 2).sum(1)).abs().max().item() <= 1e-5 * (xo.double() ** 2).sum(1).max().item()
    # two launches, same bits
    x2, s2, o2 = keep.clone(), torch.empty_like(stats), torch.empty_like(x)
    L.check(lib.esmk_gemm_residual_stats(A.data_ptr(), K, Wp.data_ptr(), M, N, K, bias.data_ptr(), 0.8, x2.data_ptr(), o2.data_ptr(),
                                         None, s2.data_ptr(), _st()))
    assert torch.equal(o2, xo) and torch.equal(s2, stats)


def _gate(out, ref, valid):
    emb = out.embeddings.cpu()[valid]
    cos = torch.nn.functional.cosine_similarity(emb, ref["embeddings"][valid], dim=-1).min().item()
    err = (out.sequence_logits.cpu()[valid] - ref["sequence_logits"][valid]).abs().max().item()
    lim = 0.05 + 0.02 * ref["sequence_logits"][valid].abs().max().item()
    assert cos >= 0.999 and err <= lim, (cos, err, lim)
    return cos, err


def test_esmc_small_path_and_large_path_against_oracle():
    """ESM C 300M-shaped (d 960, 15 heads), 6 blocks, a ragged padded batch of 700 tokens: the default engine takes the
    small-batch path, the same model with small_tokens = 0 the large-batch path; both meet the gate against the oracle,
    hidden states included, and they agree with each other far inside it."""
    from esm_amd.esmc import ESMC
    from oracle import trunk_ref  # checker only
    d, H, NL = 960, 15, 6
    w = synth.esmc_weights(d, NL, seed=21)
    tok = synth.synth_tokens([256, 131, 64, 249], seed=12)
    ref = trunk_ref.esmc_forward(tok, w, H, NL)
    valid = tok != 1
    outs = []
    for small in (1024, 0):
        m = ESMC(d, H, NL)
        m.load_state_dict(w)
        m.transformer.small_tokens = small
        m = m.cuda()
        out = m(tok.cuda())
        assert bool(m.transformer.engine()._folded) == (small > 0)
        _gate(out, ref, valid)
        hs = out.hidden_states.cpu()[:, valid]
        hr = ref["hidden_states"][:, valid]
        assert torch.nn.functional.cosine_similarity(hs, hr, dim=-1).min().item() >= 0.999
        assert out.hidden_states.cpu()[:, ~valid].abs().max().item() == 0.0
        one = m(tok.cuda(), hidden_layer=3)
        assert torch.equal(one.hidden_states[0], out.hidden_states[3])
        outs.append(out)
    a, b = outs[0].embeddings[valid.cuda()], outs[1].embeddings[valid.cuda()]
    assert torch.nn.functional.cosine_similarity(a, b, dim=-1).min().item() >= 0.9998
    # the small path is deterministic too
    again = m.__class__(d, H, NL)
    again.load_state_dict(w)
    again = again.cuda()
    o2 = again(tok.cuda())
    assert torch.equal(o2.embeddings, outs[0].embeddings) and torch.equal(o2.hidden_states, outs[0].hidden_states)


def test_esm3_small_path_with_geometric_block_against_oracle():
    """ESM3-shaped (d 512, 8 heads, 16 geometric heads, 4 blocks), one 300-residue protein with structure: the geometric
    block's s_norm is folded into its projection (fp32 output), its out-projection updates x in a residual epilogue."""
    from esm_amd.esm3 import ESM3
    from oracle import trunk_ref  # checker only
    d, H, VH, NL = 512, 8, 16, 4
    w = synth.esm3_weights(d, NL, VH, seed=5)
    Ln = 300
    seq = synth.synth_tokens([Ln], seed=3)
    g = torch.Generator().manual_seed(4)
    struct = torch.randint(0, 4096, (1, Ln), generator=g)
    coords = synth.synth_backbone(1, Ln, seed=6, nan_frac=0.1)
    coords[:, 0] = float("inf"); coords[:, Ln - 1] = float("inf")
    ref = trunk_ref.esm3_forward(w, H, VH, NL, sequence_tokens=seq, structure_tokens=struct, structure_coords=coords)
    res = []
    for small in (1024, 0):
        m = ESM3(d, H, VH, NL)
        m.load_state_dict(w)
        m.transformer.small_tokens = small
        m = m.cuda()
        out = m(sequence_tokens=seq.cuda(), structure_tokens=struct.cuda(), structure_coords=coords.cuda())
        assert bool(m.transformer.engine()._folded) == (small > 0)
        cos = torch.nn.functional.cosine_similarity(out.embeddings.cpu(), ref["embeddings"], dim=-1).min().item()
        assert cos >= 0.999, cos
        for k in ("sequence_logits", "structure_logits", "sasa_logits", "function_logits"):
            r = ref[k]
            assert (getattr(out, k).cpu() - r).abs().max().item() <= 0.05 + 0.02 * r.abs().max().item(), k
        res.append(out.embeddings)
    assert torch.nn.functional.cosine_similarity(res[0], res[1], dim=-1).min().item() >= 0.9998
