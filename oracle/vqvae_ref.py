"""CPU fp32 restatement of the VQ-VAE structure decoder's `decode`.

TEST INFRASTRUCTURE ONLY: imported by `tests/` as the checker, never by `esm_amd/`.  Written from the reference's
behaviour (citations relative to /root/reference) in plain functional torch on a flat weight dict with the
reference's state_dict keys; pinned against the imported reference by `oracle/make_golden.py vqdec_case`
(tests/golden/vqdec_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .trunk_ref import build_frames, geometric_attention, layer_norm, swiglu_ffn, transformer_stack

BB_COORDINATES = torch.tensor([[0.5256, 1.3612, 0.0], [0.0, 0.0, 0.0], [-1.5251, 0.0, 0.0]])  # constants/physics.py:1-5
SPECIAL_MIN = 4096  # min(VQVAE_SPECIAL_TOKENS.values()), constants/esm3.py:13-20
PAIRWISE_BINS = (64, 96, 64)  # distogram, direction, pae (vqvae.py:345-351)
MAX_PAE_BIN = 31.0


def _mlp(x, w, prefix, names):
    """Linear -> GELU(erf) -> LayerNorm -> Linear (vqvae.py:106-111, structure_proj.py:43-46)."""
    a, n, b = (prefix + s for s in names)
    y = F.gelu(x @ w[a + ".weight"].T + w[a + ".bias"])
    y = layer_norm(y, w[n + ".weight"], w[n + ".bias"])
    return y @ w[b + ".weight"].T + w[b + ".bias"]


def gram_schmidt(x_axis, xy_plane, eps):
    """affine3d.py:494-509: basis vectors as the COLUMNS of the rotation."""
    e0 = x_axis / torch.sqrt((x_axis ** 2).sum(-1, keepdim=True) + eps)
    e1 = xy_plane - e0 * (e0 * xy_plane).sum(-1, keepdim=True)
    e1 = e1 / torch.sqrt((e1 ** 2).sum(-1, keepdim=True) + eps)
    e2 = torch.cross(e0, e1, dim=-1)
    return torch.stack([e0, e1, e2], dim=-1)


def dim6_frames(proj, trans_scale=10.0):
    """structure_proj.py:46-63 with affine=None, affine_mask all False -> (affine [.., 12], bb [.., 3, 3])."""
    trans, x, y = proj[..., 0:3] * trans_scale, proj[..., 3:6], proj[..., 6:9]
    x = x / (x.norm(dim=-1, keepdim=True) + 1e-5)
    y = y / (y.norm(dim=-1, keepdim=True) + 1e-5)
    # Affine3D.from_graham_schmidt(neg_x_axis=x + trans, origin=trans, xy_plane=y + trans), eps 1e-10 (:417-429)
    rot = gram_schmidt(trans - (x + trans), (y + trans) - trans, 1e-10)
    bb = torch.einsum("...ij,kj->...ki", rot, BB_COORDINATES) + trans[..., None, :]
    return torch.cat([rot.flatten(-2), trans], dim=-1), bb


def pae_bins(max_bin=MAX_PAE_BIN, num_bins=64):
    """predicted_aligned_error.py:20-29."""
    step = max_bin / (num_bins - 2)
    centers = torch.linspace(0, max_bin, num_bins - 1) + step / 2
    return torch.cat([centers, centers[-1:] + step])


def pairwise_logits(x, w, prefix="pairwise_classification_head."):
    """vqvae.py:71-95 -> [B, L, L, 224]; pair (a, c) = (k_a, q_c)."""
    z = x @ w[prefix + "downproject.weight"].T
    q, k = z.chunk(2, dim=-1)
    feat = torch.cat([q[:, None, :, :] * k[:, :, None, :], q[:, None, :, :] - k[:, :, None, :]], dim=-1)
    h = F.gelu(feat @ w[prefix + "linear1.weight"].T)
    h = layer_norm(h, w[prefix + "norm.weight"], w[prefix + "norm.bias"])
    return h @ w[prefix + "linear2.weight"].T


def pae_and_ptm(pae_logits, aa_mask, max_bin=MAX_PAE_BIN):
    """predicted_aligned_error.py:37-66."""
    bins = pae_bins(max_bin, pae_logits.shape[-1])
    sq = aa_mask[:, :, None] & aa_mask[:, None, :]
    probs = pae_logits.masked_fill(~sq[..., None], torch.finfo(pae_logits.dtype).min).softmax(-1)
    pae = (probs * bins).sum(-1)
    seqlens = aa_mask.sum(-1, keepdim=True)
    d0 = 1.24 * (seqlens.clamp_min(19) - 15) ** (1 / 3) - 1.8
    f_d = 1.0 / (1 + (bins / d0.unsqueeze(-1)) ** 2)
    tm = (probs * f_d.unsqueeze(-2)).sum(-1)
    tm = (sq * tm).sum(-1) / (1e-10 + sq.sum(-1))
    return pae, tm.max(dim=-1).values


def decode(structure_tokens, w, n_heads, n_layers, sequence_id=None):
    """StructureTokenDecoder.decode (vqvae.py:365-437); the attention mask argument is unused there."""
    if sequence_id is None:
        sequence_id = torch.zeros_like(structure_tokens)
    x = w["embed.weight"][structure_tokens]
    x, _, _ = transformer_stack(x, sequence_id, w, "decoder_stack.", n_heads, n_layers, scale_residue=False)
    affine, bb = dim6_frames(_mlp(x, w, "affine_output_projection.", ("ffn1", "norm", "proj")))
    pae_logits = pairwise_logits(x, w).split(PAIRWISE_BINS, dim=-1)[2]
    pae, ptm = pae_and_ptm(pae_logits, structure_tokens < SPECIAL_MIN)
    pl = _mlp(x, w, "plddt_head.", ("dense", "norm", "output"))
    v_bins = (torch.arange(pl.shape[-1], dtype=torch.float32) + 0.5) / pl.shape[-1]  # vqvae.py:118-121
    return dict(tensor7_affine=affine, bb_pred=bb, plddt=pl.softmax(-1) @ v_bins, ptm=ptm, predicted_aligned_error=pae)


# --------------------------------------------------------------------------------------------------------------
# structure encoder (vqvae.py:179-322)
# --------------------------------------------------------------------------------------------------------------
MAX_SUPPORTED_DISTANCE = 1e6


def knn_graph(ca, coord_mask, padding_mask, sequence_id, no_knn):
    """esm/utils/misc.py:135-174 with a STABLE sort (what torch.sort does on the CPU): ties - equal sequence
    distances left and right of a residue without coordinates, +inf - keep the lower index first."""
    L = ca.shape[-2]
    pair_no_coords = ~(coord_mask[..., None, :] & coord_mask[..., :, None])
    pad = padding_mask[..., None, :] | padding_mask[..., :, None]
    pad = pad | (sequence_id.unsqueeze(1) != sequence_id.unsqueeze(2))
    dists = (ca.unsqueeze(-2) - ca.unsqueeze(-3)).norm(dim=-1)
    ar = torch.arange(L)
    seq_d = (ar[:, None] - ar[None, :]).abs().to(dists.dtype) * 1e2 + MAX_SUPPORTED_DISTANCE
    d = torch.where(pair_no_coords, seq_d, dists).masked_fill(pad, torch.inf)
    dv, edges = d.sort(dim=-1, stable=True)
    k = min(no_knn, L)
    return edges[..., :k], dv[..., :k].isfinite()


def encode(coords, w, v_heads, n_layers, attention_mask=None, sequence_id=None, residue_index=None, knn=16, bins=32):
    """StructureTokenEncoder.encode -> (z_q [B, L, d_out], indices [B, L], z [B, L, d_out] before quantisation)."""
    coords = coords[..., :3, :]
    rot, trans, amask = build_frames(coords)
    B, L = amask.shape
    if attention_mask is None:
        attention_mask = torch.ones(B, L, dtype=torch.bool)
    if sequence_id is None:
        sequence_id = torch.zeros(B, L, dtype=torch.int64)
    c = coords.clone().float()
    c[~amask] = 0
    edges, _ = knn_graph(c[..., 1, :], amask, ~attention_mask.bool(), sequence_id, knn)
    E = edges.shape[-1]
    bi = torch.arange(B)[:, None, None]
    g = lambda t: t[bi, edges].reshape(B * L, E, *t.shape[2:])  # noqa: E731  node_gather (vqvae.py:175-176)
    k_rot, k_trans, k_mask, k_seq = g(rot), g(trans), g(amask), g(sequence_id)
    res = edges.reshape(B * L, E) if residue_index is None else g(residue_index)
    diff = (res - res[:, :1]).clamp(-bins, bins) + bins + 1  # RelativePositionEmbedding (vqvae.py:45-48)
    z = w["relative_positional_embedding.embedding.weight"][diff]
    chain = torch.zeros(B * L, E, dtype=torch.int64)
    for i in range(n_layers):  # UnifiedTransformerBlock: geometric attention + FFN, scaling factor 1 (blocks.py:146-157)
        p = f"transformer.blocks.{i}."
        z = z + geometric_attention(z, k_rot, k_trans, k_mask, k_seq, chain, w, p + "geom_attn.", v_heads,
                                    mask_and_zero_frameless=False)
        z = z + swiglu_ffn(z, w, p + "ffn.")
    z = z[:, 0, :].reshape(B, L, -1).masked_fill(~amask[..., None], 0)
    z = z @ w["pre_vq_proj.weight"].T + w["pre_vq_proj.bias"]
    cb = w["codebook.embeddings"]
    flat = z.reshape(-1, z.shape[-1])
    dist = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ cb.T + (cb.T ** 2).sum(0, keepdim=True)  # codebook.py:63-67
    idx = dist.argmin(dim=1).reshape(B, L)
    return cb[idx], idx, z
