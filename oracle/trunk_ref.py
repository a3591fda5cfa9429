"""CPU fp32 restatement of the ESM C / ESM3 transformer-trunk forward.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`esm_amd/`) imports this
file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do,
and there only as the checker.

It is written from the reference's *behaviour* (file:line citations are relative to
/root/reference) in plain functional torch on a flat `dict[str, Tensor]` of weights that
uses the reference's state_dict key names.  It is pinned against the imported reference
by `oracle/make_golden.py` (fixtures under tests/golden/) - the reference's own tests
hold no numerical fixture for this path.

All math is fp32 (or whatever dtype the weights carry); no autocast, no fused kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # torch default; no LayerNorm on the path overrides it

# Token constants (esm/utils/constants/esm3.py:7-31)
SEQ_BOS, SEQ_PAD, SEQ_EOS, SEQ_CHAINBREAK, SEQ_MASK = 0, 1, 2, 31, 32
STRUCT_MASK, STRUCT_EOS, STRUCT_BOS, STRUCT_PAD, STRUCT_CHAINBREAK = 4096, 4097, 4098, 4099, 4100


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def swiglu_hidden(d_model: int, expansion_ratio: float = 8 / 3) -> int:
    """esm/layers/blocks.py:10-12 - round expansion*d up to a multiple of 256."""
    return int(((expansion_ratio * d_model) + 255) // 256 * 256)


def residue_scale(n_layers: int) -> float:
    """esm/layers/transformer_stack.py:50-52."""
    return math.sqrt(n_layers / 36)


def layer_norm(x, weight, bias=None):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + LN_EPS) * weight
    return y if bias is None else y + bias


def rotary_tables(seqlen: int, dim: int = 64, base: float = 10000.0, dtype=torch.float32):
    """esm/layers/rotary.py:136-181 - fp32 angles, tables cast to the q/k dtype."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(seqlen, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return torch.cos(freqs).to(dtype), torch.sin(freqs).to(dtype)


def apply_rotary(x, cos, sin):
    """x [B, L, H, dh]; NeoX (non-interleaved) rotation, esm/layers/rotary.py:36-65."""
    half = x.shape[-1] // 2
    x1, x2 = x[..., :half], x[..., half:]
    c = cos[: x.shape[1], None, :]
    s = sin[: x.shape[1], None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)


def attention_mask(sequence_id):
    """esm/layers/attention.py:65-72: True = attend; None = no mask."""
    if sequence_id is None:
        return None
    return sequence_id.unsqueeze(-1) == sequence_id.unsqueeze(-2)


# --------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------
def _linear(x, w, name):
    """nn.Linear with an optional bias (the bias=True block variants, blocks.py:81)."""
    y = x @ w[name + ".weight"].T
    return y + w[name + ".bias"] if name + ".bias" in w else y


def multi_head_attention(x, sequence_id, w, prefix, n_heads):
    """esm/layers/attention.py:48-82 (SDPA path).  x [B, L, d].  Linear biases and the QK-LayerNorm are used when the
    weight dict has them (bias=True / qk_layernorm=False variants: attention.py:26-36)."""
    B, L, d = x.shape
    dh = d // n_heads
    h = layer_norm(x, w[prefix + "layernorm_qkv.0.weight"], w[prefix + "layernorm_qkv.0.bias"])
    qkv = _linear(h, w, prefix + "layernorm_qkv.1")
    q, k, v = qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :]
    if prefix + "q_ln.weight" in w:  # full-width (:31-33); nn.Identity when qk_layernorm=False
        q = layer_norm(q, w[prefix + "q_ln.weight"], w.get(prefix + "q_ln.bias"))
        k = layer_norm(k, w[prefix + "k_ln.weight"], w.get(prefix + "k_ln.bias"))
    cos, sin = rotary_tables(L, dh, dtype=q.dtype)
    q = apply_rotary(q.reshape(B, L, n_heads, dh), cos, sin)
    k = apply_rotary(k.reshape(B, L, n_heads, dh), cos, sin)
    v = v.reshape(B, L, n_heads, dh)
    scores = torch.einsum("bihd,bjhd->bhij", q, k) / math.sqrt(dh)
    mask = attention_mask(sequence_id)
    if mask is not None:
        scores = scores.masked_fill(~mask[:, None], float("-inf"))
    p = torch.softmax(scores, dim=-1)
    ctx = torch.einsum("bhij,bjhd->bihd", p, v).reshape(B, L, d)
    return _linear(ctx, w, prefix + "out_proj")


def swiglu_ffn(x, w, prefix):
    """esm/layers/blocks.py:15-38."""
    h = layer_norm(x, w[prefix + "0.weight"], w[prefix + "0.bias"])
    y = _linear(h, w, prefix + "1")
    hid = y.shape[-1] // 2
    y = F.silu(y[..., :hid]) * y[..., hid:]
    return _linear(y, w, prefix + "3")


def gelu_ffn(x, w, prefix):
    """esm/layers/blocks.py:41-48."""
    h = layer_norm(x, w[prefix + "0.weight"], w[prefix + "0.bias"])
    return _linear(F.gelu(_linear(h, w, prefix + "1")), w, prefix + "3")


def build_frames(coords):
    """esm/utils/structure/affine3d.py:494-560.

    coords [B, L, 3, 3] (N, CA, C).  Returns rot [B, L, 3, 3] (basis vectors are the
    COLUMNS), trans [B, L, 3], mask [B, L] bool.
    """
    eps = 1e-10  # Affine3D.from_graham_schmidt default reaches _graham_schmidt (:417-429)
    finite = torch.isfinite(coords) & (coords < 1e6)
    mask = finite.all(dim=-1).all(dim=-1)
    c = coords.clone().float()
    c[~mask] = 0

    def gram_schmidt(bb):
        n, ca, cc = bb.unbind(dim=-2)
        x_axis = ca - cc
        xy = n - ca
        e0 = x_axis / torch.sqrt((x_axis**2).sum(-1, keepdim=True) + eps)
        e1 = xy - e0 * (e0 * xy).sum(-1, keepdim=True)
        e1 = e1 / torch.sqrt((e1**2).sum(-1, keepdim=True) + eps)
        e2 = torch.cross(e0, e1, dim=-1)
        return torch.stack([e0, e1, e2], dim=-1), ca

    avg = c.sum(1) / (mask.sum(-1)[..., None, None] + 1e-8)  # [B,3,3]
    bh_rot, bh_trans = gram_schmidt(avg)  # [B,3,3], [B,3]
    has_any = mask.any(-1)[:, None, None]
    bh_rot = torch.where(has_any, bh_rot, torch.eye(3, dtype=torch.float32).expand_as(bh_rot))
    rot, trans = gram_schmidt(c)
    rot = torch.where(mask[..., None, None], rot, bh_rot[:, None])
    trans = torch.where(mask[..., None], trans, bh_trans[:, None])
    return rot, trans, mask


def geometric_attention(s, rot, trans, affine_mask, sequence_id, chain_id, w, prefix, v_heads,
                        mask_and_zero_frameless=True):
    """esm/layers/geom_attention.py:52-149; biases of s_norm / proj / out_proj are used when the weight dict has
    them (bias=True: the VQ-VAE structure encoder, which also sets mask_and_zero_frameless=False)."""
    B, L, d = s.shape
    H = v_heads
    if sequence_id is None:
        sequence_id = torch.zeros(B, L, dtype=torch.int64)
    fmin = torch.finfo(torch.float32).min
    bias = (sequence_id.unsqueeze(-1) == sequence_id.unsqueeze(-2)).unsqueeze(1).float()
    bias = bias.masked_fill(~affine_mask[:, None, None, :], fmin)
    bias = bias.masked_fill((chain_id.unsqueeze(1) != chain_id.unsqueeze(2)).unsqueeze(1), fmin)

    ns = layer_norm(s, w[prefix + "s_norm.weight"], w.get(prefix + "s_norm.bias"))
    proj = _linear(ns, w, prefix + "proj")  # [B, L, 15H]
    vec_rot = proj[..., : 9 * H].reshape(B, L, 3 * H, 3)
    vec_dist = proj[..., 9 * H :].reshape(B, L, 2 * H, 3)
    # R . p for every 3-vector (RotationMatrix.apply, affine3d.py:168-175)
    vec_rot = torch.einsum("blij,blhj->blhi", rot, vec_rot)
    vec_dist = torch.einsum("blij,blhj->blhi", rot, vec_dist) + trans[:, :, None, :]
    q_rot, k_rot, val = vec_rot[:, :, :H], vec_rot[:, :, H : 2 * H], vec_rot[:, :, 2 * H :]
    q_dist, k_dist = vec_dist[:, :, :H], vec_dist[:, :, H:]

    dist = (q_dist.permute(0, 2, 1, 3)[:, :, :, None, :] - k_dist.permute(0, 2, 1, 3)[:, :, None, :, :]).norm(dim=-1)
    dist = dist / math.sqrt(3)
    rterm = torch.einsum("bihc,bjhc->bhij", q_rot, k_rot) / math.sqrt(3)
    w_d = F.softplus(w[prefix + "distance_scale_per_head"])[None, :, None, None]
    w_r = F.softplus(w[prefix + "rotation_scale_per_head"])[None, :, None, None]
    a = rterm * w_r - dist * w_d + bias
    p = torch.softmax(a, dim=-1)
    o = torch.einsum("bhij,bjhc->bihc", p, val)  # [B, L, H, 3] global frame
    o = torch.einsum("blji,blhj->blhi", rot, o)  # R^T . o
    o = o.reshape(B, L, 3 * H)
    if mask_and_zero_frameless:
        o = o.masked_fill(~affine_mask[..., None], 0.0)
    return _linear(o, w, prefix + "out_proj")


def transformer_block(x, sequence_id, frames, chain_id, w, prefix, n_heads, v_heads, scale, geom,
                      ffn_type="swiglu", mask_and_zero_frameless=True):
    """esm/layers/blocks.py:146-157 (ffn_type: blocks.py:109-114)."""
    x = x + multi_head_attention(x, sequence_id, w, prefix + "attn.", n_heads) / scale
    if geom:
        rot, trans, fmask = frames
        x = x + geometric_attention(x, rot, trans, fmask, sequence_id, chain_id, w, prefix + "geom_attn.", v_heads,
                                    mask_and_zero_frameless=mask_and_zero_frameless) / scale
    ffn = swiglu_ffn if ffn_type == "swiglu" else gelu_ffn
    x = x + ffn(x, w, prefix + "ffn.") / scale
    return x


def transformer_stack(x, sequence_id, w, prefix, n_heads, n_layers, v_heads=None, n_layers_geom=0,
                      frames=None, chain_id=None, scale_residue=True, ffn_type="swiglu", mask_and_zero_frameless=True):
    """esm/layers/transformer_stack.py:64-94 -> (post_norm, pre_norm, hiddens).  scale_residue=False is the
    VQ-VAE structure decoder's stack (esm/models/vqvae.py:337-339).  bias=True / qk_layernorm=False are read off
    the weight dict (keys present or not); ffn_type and mask_and_zero_frameless are constructor options
    (transformer_stack.py:34-37) and passed in."""
    if chain_id is None:
        chain_id = torch.ones(x.shape[:-1], dtype=torch.int64)
    scale = residue_scale(n_layers) if scale_residue else 1.0
    hiddens = []
    for i in range(n_layers):
        x = transformer_block(x, sequence_id, frames, chain_id, w, f"{prefix}blocks.{i}.", n_heads,
                              v_heads, scale, geom=i < n_layers_geom, ffn_type=ffn_type,
                              mask_and_zero_frameless=mask_and_zero_frameless)
        hiddens.append(x)
    return layer_norm(x, w[prefix + "norm.weight"]), x, hiddens


def regression_head(x, w, prefix):
    """esm/layers/regression_head.py:17-22."""
    y = x @ w[prefix + "0.weight"].T + w[prefix + "0.bias"]
    y = F.gelu(y)  # erf form
    y = layer_norm(y, w[prefix + "2.weight"], w[prefix + "2.bias"])
    return y @ w[prefix + "3.weight"].T + w[prefix + "3.bias"]


# --------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------
def esmc_forward(tokens, w, n_heads, n_layers, sequence_id=None):
    """esm/models/esmc.py:117-173 (non-flash path).

    Returns dict(sequence_logits [B,L,64], embeddings [B,L,d] POST-norm,
    hidden_states [n_layers,B,L,d]).
    """
    if sequence_id is None:
        sequence_id = tokens != SEQ_PAD
    x = w["embed.weight"][tokens]
    post, _pre, hiddens = transformer_stack(x, sequence_id, w, "transformer.", n_heads, n_layers)
    return {
        "sequence_logits": regression_head(post, w, "sequence_head."),
        "embeddings": post,
        "hidden_states": torch.stack(hiddens, 0),
    }


def rbf16(values):
    """esm/utils/misc.py:105-115 with v_min=0, v_max=1, n_bins=16."""
    centers = torch.linspace(0.0, 1.0, 16, dtype=values.dtype)
    z = (values.unsqueeze(-1) - centers) / (1.0 / 16)
    return torch.exp(-(z**2))


def esm3_encode_inputs(seq, struct, avg_plddt, res_plddt, ss8, sasa, func, resid, w, prefix="encoder."):
    """esm/models/esm3.py:92-147."""
    x = w[prefix + "sequence_embed.weight"][seq]
    x = x + rbf16(avg_plddt) @ w[prefix + "plddt_projection.weight"].T + w[prefix + "plddt_projection.bias"]
    x = x + rbf16(res_plddt) @ w[prefix + "structure_per_res_plddt_projection.weight"].T \
        + w[prefix + "structure_per_res_plddt_projection.bias"]
    x = x + w[prefix + "structure_tokens_embed.weight"][struct]
    x = x + w[prefix + "ss8_embed.weight"][ss8]
    x = x + w[prefix + "sasa_embed.weight"][sasa]
    fparts = []
    for i in range(8):
        t = func[..., i]
        e = w[f"{prefix}function_embed.{i}.weight"][t]
        fparts.append(e * (t != 0)[..., None])  # padding_idx=0 contributes nothing
    x = x + torch.cat(fparts, dim=-1)
    rw = w[prefix + "residue_embed.weight"]
    x = x + (rw[resid] * (resid != 0)[..., None]).sum(dim=-2)  # EmbeddingBag(sum, padding_idx=0)
    return x


ESM3_HEADS = (("sequence", 64), ("structure", 4096), ("ss8", 11), ("sasa", 19), ("function", 260 * 8),
              ("residue", 1478))


def esm3_forward(w, n_heads, v_heads, n_layers, *, sequence_tokens=None, structure_tokens=None,
                 ss8_tokens=None, sasa_tokens=None, function_tokens=None, residue_annotation_tokens=None,
                 average_plddt=None, per_res_plddt=None, structure_coords=None, chain_id=None,
                 sequence_id=None):
    """esm/models/esm3.py:260-374.  embeddings = PRE-norm."""
    given = [t for t in (sequence_tokens, structure_tokens, ss8_tokens, sasa_tokens, structure_coords,
                         function_tokens, residue_annotation_tokens) if t is not None]
    if not given:
        raise ValueError("At least one of the inputs must be non-None")
    L = given[0].shape[1]

    def dflt(x, tok):
        return torch.full((1, L), tok, dtype=torch.long) if x is None else x

    sequence_tokens = dflt(sequence_tokens, SEQ_MASK)
    ss8_tokens = dflt(ss8_tokens, 0)
    sasa_tokens = dflt(sasa_tokens, 0)
    average_plddt = dflt(average_plddt, 1).float()
    per_res_plddt = dflt(per_res_plddt, 0).float()
    chain_id = dflt(chain_id, 0)
    if residue_annotation_tokens is None:
        residue_annotation_tokens = torch.zeros((1, L, 16), dtype=torch.long)
    if function_tokens is None:
        function_tokens = torch.zeros((1, L, 8), dtype=torch.long)
    if structure_coords is None:
        structure_coords = torch.full((1, L, 3, 3), float("nan"))
    structure_coords = structure_coords[..., :3, :]
    frames = build_frames(structure_coords)
    structure_tokens = dflt(structure_tokens, STRUCT_MASK)
    st = structure_tokens.masked_fill(structure_tokens == -1, STRUCT_MASK)
    st = st.masked_fill(sequence_tokens == SEQ_BOS, STRUCT_BOS)
    st = st.masked_fill(sequence_tokens == SEQ_PAD, STRUCT_PAD)
    st = st.masked_fill(sequence_tokens == SEQ_EOS, STRUCT_EOS)
    st = st.masked_fill(sequence_tokens == SEQ_CHAINBREAK, STRUCT_CHAINBREAK)

    x = esm3_encode_inputs(sequence_tokens, st, average_plddt, per_res_plddt, ss8_tokens, sasa_tokens,
                           function_tokens, residue_annotation_tokens, w)
    post, pre, _ = transformer_stack(x, sequence_id, w, "transformer.", n_heads, n_layers, v_heads=v_heads,
                                     n_layers_geom=1, frames=frames, chain_id=chain_id)
    out = {}
    for name, _dim in ESM3_HEADS:
        out[name] = regression_head(post, w, f"output_heads.{name}_head.")
    B = post.shape[0]
    return {
        "sequence_logits": out["sequence"],
        "structure_logits": out["structure"],
        "secondary_structure_logits": out["ss8"],
        "sasa_logits": out["sasa"],
        "function_logits": out["function"].reshape(B, L, 8, 260),
        "residue_logits": out["residue"],
        "embeddings": pre,
    }
