"""The q side of QK-LayerNorm + rotary inside the attention kernel (esmk_gemm_qkv_stats + esmk_k_prep +
esmk_attn_varlen_qfused; reference attention.py:31-33,51-55, rotary.py:47-65): against a torch fp32 computation of the
same op, against the three-call path (esmk_gemm + esmk_qk_prep_ex + esmk_attn_varlen) it replaces, and at model level
(the trunk takes the fused path whenever esmk_qkv_stats_parts() > 0)."""
import math

import numpy as np
import pytest
import torch

from esm_amd import synth

pytestmark = pytest.mark.gpu


def _s():
    return torch.cuda.current_stream().cuda_stream


def _reference(qkv, cu, H, gq, gk, bq, bk, positions):
    from oracle import trunk_ref
    N, d3 = qkv.shape
    d = d3 // 3
    q = trunk_ref.layer_norm(qkv[:, :d], gq)
    k = trunk_ref.layer_norm(qkv[:, d:2 * d], gk)
    if bq is not None:
        q, k = q + bq, k + bk
    cos, sin = trunk_ref.rotary_tables(int(positions.max()) + 1)
    c, s = cos[positions][:, None, :], sin[positions][:, None, :]

    def rot(x):
        x = x.reshape(N, H, 64)
        x1, x2 = x[..., :32], x[..., 32:]
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)

    q, k, v = rot(q), rot(k), qkv[:, 2 * d:].reshape(N, H, 64)
    out = torch.zeros(N, H, 64)
    for a, b in zip(cu[:-1], cu[1:]):
        sc = torch.einsum("ihd,jhd->hij", q[a:b], k[a:b]) / 8.0
        out[a:b] = torch.einsum("hij,jhd->ihd", torch.softmax(sc, -1), v[a:b])
    return out.reshape(N, d)


@pytest.mark.parametrize("lens,H,biased", [([512] * 9 + [300, 129, 64, 1], 18, False), ([700, 1300, 64, 1, 800], 18, True),
                                           ([512] * 15 + [400, 112], 24, False),
                                           # d 960 / 2560: the 256-tile kernel's statistics epilogue (64- / 128-channel parts;
                                           # d 960: the q / k boundary lies inside a tile, and the last tile column splits off)
                                           ([512] * 8 + [300, 129, 1], 15, False), ([700, 1300, 64, 1, 800, 900], 15, True),
                                           ([512] * 3 + [300, 129, 1], 40, False)])
def test_qfused_against_reference_and_three_call_path(lens, H, biased):
    from esm_amd import _lib as L
    from esm_amd import engine
    from esm_amd.packing import pack_lengths
    lib = L.lib()
    d = 64 * H
    pb = pack_lengths(lens)
    M = pb.n_tokens
    parts = lib.esmk_qkv_stats_parts(M, d)
    # d 960: 4526 tokens take 256-wide tiles (64-channel parts), 3765 tokens 192-wide ones (one part per wave half-tile)
    # d 1152: 5102 tokens take 288-wide tiles, 2865 tokens 192-wide ones (12 parts)
    assert parts == {18: 8 if M > 4352 else 12, 24: 16, 15: 15 if M > 4352 else 10, 40: 20}[H], parts
    g = torch.Generator().manual_seed(M + H)
    xn = torch.randn(M, d, generator=g).bfloat16()
    W = torch.randn(3 * d, d, generator=g) * 0.04 + 0.004   # non-zero row means: q and k are not centred
    gq, gk = 1 + 0.1 * torch.randn(d, generator=g), 1 + 0.1 * torch.randn(d, generator=g)
    bq = 0.1 * torch.randn(d, generator=g) if biased else None
    bk = 0.1 * torch.randn(d, generator=g) if biased else None
    pos = torch.from_numpy(pb.positions.astype(np.int64))
    qkv_ref = xn.float() @ W.bfloat16().float().T
    ref = _reference(qkv_ref, pb.cu_seqlens.tolist(), H, gq, gk, bq, bk, pos)

    cos, sin = engine.rotary_tables(int(pos.max()) + 1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
    positions, cu, qmap = dev(pb.positions), dev(pb.cu_seqlens), dev(pb.qblk_map.reshape(-1))
    cosd, sind, gqd, gkd = cos.cuda(), sin.cuda(), gq.cuda(), gk.cuda()
    bqd, bkd = (bq.cuda(), bk.cuda()) if biased else (None, None)
    ptr = lambda t: t.data_ptr() if t is not None else None
    q_scale = 0.125 * math.log2(math.e)
    xd, Wp = xn.cuda(), engine.pack_weight(W, "cuda")
    n_qblk = pb.qblk_map.shape[0]

    # three-call path
    qkv_a = torch.empty(M, 3 * d, dtype=torch.bfloat16, device="cuda")
    L.check(lib.esmk_gemm(xd.data_ptr(), d, Wp.data_ptr(), M, 3 * d, d, L.EPI_BF16, None, 1.0, qkv_a.data_ptr(), _s()))
    raw = qkv_a.clone()
    L.check(lib.esmk_qk_prep_ex(qkv_a.data_ptr(), gqd.data_ptr(), gkd.data_ptr(), ptr(bqd), ptr(bkd), positions.data_ptr(),
                                cosd.data_ptr(), sind.data_ptr(), cos.shape[0], M, d, H, q_scale, 1e-5, _s()))
    ctx_a = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.esmk_attn_varlen(qkv_a.data_ptr(), ctx_a.data_ptr(), cu.data_ptr(), qmap.data_ptr(), n_qblk, M, d, H, _s()))

    # fused path
    qkv_b = torch.empty(M, 3 * d, dtype=torch.bfloat16, device="cuda")
    stats = torch.full((M, parts, 2), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.esmk_gemm_qkv_stats(xd.data_ptr(), d, Wp.data_ptr(), M, d, qkv_b.data_ptr(), stats.data_ptr(), _s()))
    torch.cuda.synchronize()
    # the projection itself is the same arithmetic (every tile kernel adds the same MFMA products in the same K order)
    assert torch.equal(qkv_b, raw)
    qr = qkv_ref[:, :d].reshape(M, parts, d // parts)
    want = torch.stack([qr.sum(-1), (qr * qr).sum(-1)], -1)
    got = stats.cpu()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item()
    qmr = torch.full((M, 2), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.esmk_k_prep(qkv_b.data_ptr(), gkd.data_ptr(), ptr(bkd), positions.data_ptr(), cosd.data_ptr(), sind.data_ptr(),
                            cos.shape[0], M, d, H, 1e-5, stats.data_ptr(), parts, qmr.data_ptr(), _s()))
    torch.cuda.synchronize()
    q32 = qkv_ref[:, :d]
    want_mr = torch.stack([q32.mean(-1), (q32.var(-1, unbiased=False) + 1e-5).rsqrt()], -1)
    assert (qmr.cpu() - want_mr).abs().max().item() <= 1e-3 * want_mr.abs().max().item()
    assert torch.equal(qkv_b[:, :d], raw[:, :d]) and torch.equal(qkv_b[:, 2 * d:], raw[:, 2 * d:])   # q and v untouched
    ka, kb = qkv_a[:, d:2 * d].float(), qkv_b[:, d:2 * d].float()
    assert (ka - kb).abs().max().item() <= 0.01 * ka.abs().max().item()   # same op, other summation order: <= 1 bf16 ulp
    ctx_b = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.esmk_attn_varlen_qfused(qkv_b.data_ptr(), ctx_b.data_ptr(), cu.data_ptr(), qmap.data_ptr(), n_qblk, M, d, H, 0,
                                        qmr.data_ptr(), gqd.data_ptr(), ptr(bqd), positions.data_ptr(),
                                        cosd.data_ptr(), sind.data_ptr(), cos.shape[0], q_scale, _s()))
    torch.cuda.synchronize()
    a, b = ctx_a.float().cpu(), ctx_b.float().cpu()
    assert torch.isfinite(b).all()
    scale = max(1.0, ref.abs().max().item())
    assert (b - ref).abs().max().item() <= 0.03 * scale
    assert (a - ref).abs().max().item() <= 0.03 * scale
    assert (a - b).abs().max().item() <= 0.02 * scale
    # the fused path is not further from the fp32 result than the three-call path (statistics come from fp32 accumulators)
    assert (b - ref).pow(2).mean().sqrt().item() <= 1.1 * (a - ref).pow(2).mean().sqrt().item() + 1e-4


@pytest.mark.parametrize("M,H", [(1, 2), (7, 15), (300, 18), (129, 40)])
def test_k_prep_alone_matches_qk_prep_on_k(M, H):
    """esmk_k_prep without q statistics (qstats NULL) on any head count / token count, odd tails included: the k
    columns equal esmk_qk_prep's, q and v are untouched."""
    from esm_amd import _lib as L
    from esm_amd import engine
    lib = L.lib()
    d = 64 * H
    g = torch.Generator().manual_seed(M * 31 + H)
    qkv = (torch.randn(M, 3 * d, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    gq, gk = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (1 + 0.1 * torch.randn(d, generator=g)).cuda()
    pos = torch.randint(0, 700, (M,), generator=g).int().cuda()
    cos, sin = [t.cuda() for t in engine.rotary_tables(700)]
    a, b = qkv.clone(), qkv.clone()
    L.check(lib.esmk_qk_prep(a.data_ptr(), gq.data_ptr(), gk.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), 700, M, d, H,
                             0.18, 1e-5, _s()))
    L.check(lib.esmk_k_prep(b.data_ptr(), gk.data_ptr(), None, pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), 700, M, d, H, 1e-5,
                            None, 0, None, _s()))
    torch.cuda.synchronize()
    assert torch.equal(b[:, :d], qkv[:, :d]) and torch.equal(b[:, 2 * d:], qkv[:, 2 * d:])
    ka, kb = a[:, d:2 * d].float(), b[:, d:2 * d].float()
    assert torch.isfinite(kb).all()
    assert (ka - kb).abs().max().item() <= 0.01 * ka.abs().max().item()


def test_qfused_rejects_bad_arguments():
    from esm_amd import _lib as L
    lib = L.lib()
    assert lib.esmk_qkv_stats_parts(1024, 1152) == 0      # too few tiles: the 128-tile kernel runs
    assert lib.esmk_qkv_stats_parts(2048, 1152) == 12     # 192-channel tiles
    assert lib.esmk_qkv_stats_parts(32768, 2560) == 20    # 256-channel tiles, 128-channel parts
    assert lib.esmk_qkv_stats_parts(32768, 960) == 0      # 288-channel tiles straddle the q / k boundary (and beat narrower ones)
    assert lib.esmk_qkv_stats_parts(4526, 960) == 15      # 256-channel tiles, 64-channel parts
    assert lib.esmk_qkv_stats_parts(8192, 960) == 10      # 192-channel tiles
    assert lib.esmk_qkv_stats_parts(2048, 960) == 0       # too few tiles
    assert lib.esmk_qkv_stats_parts(32768, 1000) == 0     # d % 64
    assert lib.esmk_qkv_stats_parts(32768, 1152) == 8
    x = torch.zeros(8, device="cuda")
    rc = lib.esmk_gemm_qkv_stats(x.data_ptr(), 1152, x.data_ptr(), 1024, 1152, x.data_ptr(), x.data_ptr(), _s())
    assert rc == -2 and b"esmk_qkv_stats_parts" in lib.esmk_last_error()
    rc = lib.esmk_attn_varlen_qfused(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 64, 128, 2, 0, None,
                                     x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, 1.0, _s())
    assert rc == -1
    rc = lib.esmk_k_prep(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, 64, 128, 2, 1e-5,
                         x.data_ptr(), 3, x.data_ptr(), _s())
    assert rc == -1   # parts must divide d


@pytest.mark.parametrize("d,H,L_,lens,parts", [
    (1152, 18, 512, [512, 30, 20, 100, 1, 64, 50, 28], 18),          # 4096 positions: 256-wide tiles, 64-channel parts
    (1152, 18, 512, [512, 30, 20, 100, 1, 64, 50, 28, 7, 30] + [20] * 10, 8),  # 10240 positions: 288-wide tiles
    (960, 15, 512, [512, 30, 20, 100, 1, 64, 50, 28] * 2, 10),       # ESM C 300M width, 8192 positions: 192-wide tiles
    (2560, 40, 256, [256, 30, 20, 100, 1, 64, 50, 28], 20)])         # ESM C 6B width: 256-wide tiles, 128-channel parts
def test_trunk_fused_and_unfused_paths_agree_on_a_padded_batch(d, H, L_, lens, parts):
    """Two blocks of every ESM C width: the default forward of a padded batch packs all B x L positions (fused path,
    pad_tail layout); drop_pads=True packs only the valid ones on the host (three-call path: too few tiles)."""
    from esm_amd import _lib as L
    from esm_amd.esmc import ESMC
    m = ESMC(d, H, 2)
    m.load_state_dict(synth.esmc_weights(d, 2, seed=3))
    m = m.cuda()
    assert L.lib().esmk_qkv_stats_parts(len(lens) * L_, d) == parts and L.lib().esmk_qkv_stats_parts(sum(lens), d) == 0
    tok = synth.synth_tokens(lens, pad_to=L_, seed=5).cuda()
    fused = m(tok)
    plain = m(tok, drop_pads=True)
    torch.cuda.synchronize()
    valid = (tok != 1).cpu()
    for name in ("embeddings", "sequence_logits"):
        a, b = getattr(fused, name).cpu()[valid], getattr(plain, name).cpu()[valid]
        assert torch.isfinite(a).all()
        # two valid kernel paths (other GEMM tilings, other statistics source) through two bf16 blocks and the head
        assert (a - b).abs().max().item() <= 0.02 * max(1.0, b.abs().max().item()), name
        cosv = torch.nn.functional.cosine_similarity(a, b, dim=-1).min().item()
        assert cosv >= 0.9995, (name, cosv)
    assert fused.embeddings.cpu()[~valid].abs().max().item() == 0.0
