"""Function-token decoder forward on the HIP path (esm_amd/function_decoder.py, csrc/attn128.hip, esmk_head_forward_ex)
through the C-ABI, against outputs of the reference's own forward (tests/golden/fdec_*.npz; the reference module is
assembled without its data-file-reading constructor, oracle/ref_import.py build_fdec).

Gate: every logit within 1.5 x the drift of the reference's OWN bf16 (autocast) run on the same inputs, stored in the
fixture (same arithmetic class: bf16 operands, fp32 accumulation), i.e. about 1 % of the largest logit.
"""
import numpy as np
import pytest
import torch

from tests import _golden as G

pytestmark = pytest.mark.gpu
HEADS = ("keyword_logits", "keyword_tfidf", "interpro_logits")


def _model(fx):
    from esm_amd.function_decoder import FunctionTokenDecoder, FunctionTokenDecoderConfig
    cfg = FunctionTokenDecoderConfig(d_model=int(fx["d"]), n_heads=int(fx["H"]), n_layers=int(fx["NL"]),
                                     num_interpro_classes=int(fx["n_ip"]), keyword_vocabulary_size=int(fx["n_kw"]))
    m = FunctionTokenDecoder(cfg)
    m.load_state_dict(G.weights_for(fx), strict=True)
    return m.cuda()


def test_attention_l64_d128_kernel():
    """esmk_attn_l64_d128 against torch: rotary over the 128-wide head + unmasked softmax attention."""
    from esm_amd import _lib as L
    from esm_amd.engine import _stream, rotary_tables
    from oracle import trunk_ref
    g = torch.Generator().manual_seed(1)
    S, H = 5, 3
    d = 128 * H
    qkv = torch.randn(S, 64, 3 * d, generator=g).bfloat16()
    q, k, v = (qkv.float()[..., i * d:(i + 1) * d].reshape(S, 64, H, 128) for i in range(3))
    cos, sin = rotary_tables(64, dim=128)
    q, k = trunk_ref.apply_rotary(q, cos, sin), trunk_ref.apply_rotary(k, cos, sin)
    p = torch.softmax(torch.einsum("bihd,bjhd->bhij", q, k) / 128 ** 0.5, dim=-1)
    want = torch.einsum("bhij,bjhd->bihd", p, v).reshape(S * 64, d)
    qd = qkv.reshape(S * 64, 3 * d).contiguous().cuda()
    ctx = torch.empty(S * 64, d, dtype=torch.bfloat16, device="cuda")
    cd, sd = cos.cuda(), sin.cuda()
    L.check(L.lib().esmk_attn_l64_d128(L.ptr(qd), L.ptr(ctx), L.ptr(cd), L.ptr(sd), S, d, H, _stream()))
    np.testing.assert_allclose(ctx.float().cpu().numpy(), want.numpy(), atol=2e-2, rtol=0)
    with pytest.raises(L.EsmkError):
        L.check(L.lib().esmk_attn_l64_d128(L.ptr(qd), L.ptr(ctx), L.ptr(cd), L.ptr(sd), S, d, H + 1, _stream()))


@pytest.mark.parametrize("name", ["fdec_tiny", "fdec_d1024_3l"])
def test_forward_matches_reference(name):
    fx = G.load(name)
    m = _model(fx)
    out = m(G.t(fx["tokens"]).cuda())
    st = int(fx["stride"])
    for k in HEADS:
        got = out[k].cpu().numpy()
        assert got.dtype == np.float32 and got.shape[0] == fx["tokens"].shape[0]
        err = float(np.abs(got[:, ::st] - fx[k]).max())
        gate = 1.5 * float(fx["ref_bf16_maxdiff_" + k])
        assert err <= gate, (k, err, gate)
    # rows made of special tokens collapse to the same input (function_decoder.py:161-163) -> identical outputs
    np.testing.assert_array_equal(out["interpro_logits"][1].cpu().numpy(), out["interpro_logits"][2].cpu().numpy())


def test_contract_and_batch_independence():
    fx = G.load("fdec_tiny")
    m = _model(fx)
    g = torch.Generator().manual_seed(2)
    tok = torch.randint(4, 260, (300, 8), generator=g).cuda()  # one row per residue of a 300-residue protein
    a = m(tok)
    b = m(tok[100:140])
    for k in HEADS:
        assert a[k].shape == (300, int(fx["n_kw"] if k != "interpro_logits" else fx["n_ip"]))
        assert torch.isfinite(a[k]).all()
        assert float((a[k][100:140] - b[k]).abs().max()) <= 0.05  # rows are independent (GEMM tiling differs with M)
    with pytest.raises(NotImplementedError):
        m.decode(tok, None)
    with pytest.raises(AssertionError):
        m(tok[:, :5])
