"""The CPU oracle (oracle/trunk_ref.py) against fixtures produced by the imported reference."""
import pytest
import torch

from oracle import function_decoder_ref, trunk_ref, vqvae_ref
from tests import _golden as G

TOL = 2e-5  # fp32 vs fp32, different op order


def _close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("name", ["esmc_tiny_bool", "esmc_tiny_binpacked", "esmc_300m_2x128", "esmc_600m_2x128"])
def test_esmc(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    tok = G.t(fx["tokens"])
    sid = G.t(fx["sequence_id"]) if "sequence_id" in fx else None
    out = trunk_ref.esmc_forward(tok, w, int(fx["H"]), int(fx["NL"]), sequence_id=sid)
    s = int(fx["token_stride"])
    valid = (tok != 1)[:, ::s]
    _close(out["sequence_logits"][:, ::s][valid], G.t(fx["sequence_logits"])[valid])
    _close(out["embeddings"][:, ::s][valid], G.t(fx["embeddings"])[valid])
    hl = [int(i) for i in fx["hidden_layers"]]
    _close(out["hidden_states"][hl][:, :, ::s][:, valid], G.t(fx["hidden_states"])[:, valid])


@pytest.mark.parametrize("name", ["stack_tiny_nomask", "stack_d960", "stack_d1152", "stack_d1536", "stack_d2560", "stack_d1280_vqdec",
                                  "stack_d2560_L4096"])
def test_stack(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    sid = G.t(fx["sequence_id"]) if "sequence_id" in fx else None
    post, pre, _ = trunk_ref.transformer_stack(G.seeded(fx, "x"), sid, w, "", int(fx["H"]), int(fx["NL"]),
                                               scale_residue=bool(int(fx["scale_residue"])) if "scale_residue" in fx else True)
    s = int(fx["token_stride"])
    valid = sid[:, ::s] if sid is not None else torch.ones(post[:, ::s].shape[:2], dtype=torch.bool)
    _close(post[:, ::s][valid], G.t(fx["post"])[valid])
    _close(pre[:, ::s][valid], G.t(fx["pre"])[valid])


STACK_VARIANTS = ["stackvar_bias_noqkln_gelu", "stackvar_noqkln_swiglu", "stackvar_gelu_noscale", "stackvar_bias_qkln", "stackvar_geom_nozero",
                  "stackvar_bias_geom2"]


@pytest.mark.parametrize("name", STACK_VARIANTS)
def test_stack_variants(name):
    """The other block variants of blocks.py:73-115 through the reference's generic TransformerStack."""
    fx = G.load(name)
    w = G.weights_for(fx)
    kw = G.variant_ctor(fx)
    frames = chain = None
    ng = kw.get("n_layers_geom", 1)
    if ng > 0:
        frames = trunk_ref.build_frames(G.t(fx["coords"]))
        chain = G.t(fx["chain_id"])
    post, pre, hid = trunk_ref.transformer_stack(
        G.t(fx["x"]), G.t(fx["sequence_id"]), w, "", int(fx["H"]), int(fx["NL"]), v_heads=int(fx["VH"]) or None,
        n_layers_geom=ng, frames=frames, chain_id=chain, scale_residue=kw.get("scale_residue", True),
        ffn_type=kw.get("ffn_type", "swiglu"), mask_and_zero_frameless=kw.get("mask_and_zero_frameless", False))
    _close(post, G.t(fx["post"]), 5e-5)
    _close(pre, G.t(fx["pre"]), 5e-5)
    _close(hid[0], G.t(fx["hidden0"]), 5e-5)


def test_frames():
    fx = G.load("frames")
    rot, trans, mask = trunk_ref.build_frames(G.t(fx["coords"]))
    aff = G.t(fx["affine"])
    assert torch.equal(mask, G.t(fx["mask"]))
    _close(rot.flatten(-2), aff[..., :9], 1e-6)
    _close(trans, aff[..., 9:], 1e-6)


@pytest.mark.parametrize("name", ["geom_plain", "geom_nostruct", "geom_multichain", "geom_binpacked",
                                  "geom_onerow_frameless", "geom_d1536_vh256", "geom_d1536_vh256_L512"])
def test_geom(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    rot, trans, mask = trunk_ref.build_frames(G.t(fx["coords"]))
    sid = G.t(fx["sequence_id"]) if "sequence_id" in fx else None
    out = trunk_ref.geometric_attention(G.seeded(fx, "s"), rot, trans, mask, sid, G.t(fx["chain_id"]), w,
                                        "blocks.0.geom_attn.", int(fx["VH"]))
    rs = int(fx["row_stride"]) if "row_stride" in fx else 1
    _close(out[:, ::rs], G.t(fx["out"]))
    if name == "geom_nostruct":
        assert out.abs().max().item() == 0.0


@pytest.mark.parametrize("name", ["esm3_tiny_alltracks", "esm3_tiny_seqstruct", "esm3_tiny_chain_seqid",
                                  "esm3_d1536_2layer", "esm3_open_48l_L512"])
def test_esm3(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    out = trunk_ref.esm3_forward(w, int(fx["H"]), int(fx["VH"]), int(fx["NL"]), **G.esm3_kwargs(fx))
    bs = int(fx["big_stride"])
    es = int(fx["emb_stride"]) if "emb_stride" in fx else 1
    tol = 2e-4 if int(fx["NL"]) >= 30 else None  # fp32 summation-order drift over 48 layers
    for k in ("sequence_logits", "secondary_structure_logits", "sasa_logits"):
        _close(out[k], G.t(fx[k]), *([tol] if tol else []))
    _close(out["embeddings"][:, ::es], G.t(fx["embeddings"]), *([tol] if tol else []))
    for k in ("structure_logits", "function_logits", "residue_logits"):
        _close(out[k][:, ::bs], G.t(fx[k]), *([tol] if tol else []))


VQDEC_KEYS = ("tensor7_affine", "bb_pred", "plddt", "ptm", "predicted_aligned_error")


@pytest.mark.parametrize("name", ["vqdec_tiny", "vqdec_tiny_seqid", "vqdec_d1280_30l", "vqdec_d1280_30l_L512"])
def test_vqvae_structure_decoder(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    sid = G.t(fx["sequence_id"]) if "sequence_id" in fx else None
    out = vqvae_ref.decode(G.t(fx["tokens"]), w, int(fx["H"]), int(fx["NL"]), sequence_id=sid)
    st = int(fx["pae_stride"]) if "pae_stride" in fx else 1
    out["predicted_aligned_error"] = out["predicted_aligned_error"][:, ::st, ::st]
    for k in VQDEC_KEYS:
        _close(out[k], G.t(fx[k]), 1e-4 if name.startswith("vqdec_d1280") else TOL)


VQENC = ["vqenc_tiny", "vqenc_tiny_seqid_resindex", "vqenc_tiny_short", "vqenc_tiny_sparse", "vqenc_d1024_vh128"]


@pytest.mark.parametrize("name", VQENC)
def test_vqvae_structure_encoder(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    opt = lambda k: G.t(fx[k]) if k in fx else None  # noqa: E731
    coords, amask = G.t(fx["coords"]), G.t(fx["affine_mask"])
    sid = opt("sequence_id")
    # the kNN graph alone (find_knn_edges)
    c = coords.clone()
    c[~amask] = 0
    edges, emask = vqvae_ref.knn_graph(c[..., 1, :], amask, ~G.t(fx["attention_mask"]),
                                       sid if sid is not None else torch.zeros(amask.shape, dtype=torch.int64), 16)
    G.assert_knn_equal(edges, emask, fx)
    zq, idx, z = vqvae_ref.encode(coords, w, int(fx["VH"]), int(fx["NL"]), attention_mask=G.t(fx["attention_mask"]),
                                  sequence_id=sid, residue_index=opt("residue_index"))
    _close(z, G.t(fx["z"]), 1e-4)
    assert torch.equal(idx, G.t(fx["indices"]))
    _close(zq, G.t(fx["z_q"]), 1e-6)


@pytest.mark.parametrize("name", ["fdec_tiny", "fdec_d1024_3l"])
def test_function_token_decoder(name):
    fx = G.load(name)
    w = G.weights_for(fx)
    out = function_decoder_ref.forward(G.t(fx["tokens"]), w, int(fx["H"]), int(fx["NL"]))
    st = int(fx["stride"])
    for k in ("keyword_logits", "keyword_tfidf", "interpro_logits"):
        _close(out[k][:, ::st], G.t(fx[k]), 1e-4)
