"""VQ-VAE structure tokenizer models.

StructureTokenDecoder: ESM3 structure tokens -> backbone coordinates, pLDDT, pTM, predicted aligned error
(reference esm/models/vqvae.py:325-437, esm/layers/structure_proj.py,
esm/utils/structure/predicted_aligned_error.py, caller esm/utils/decoding.py:138-169).
StructureTokenEncoder: backbone coordinates -> structure tokens (vqvae.py:179-322, esm/utils/misc.py:135-174,
esm/layers/codebook.py:57-84, caller esm/utils/encoding.py tokenize_structure); see the class docstring.

Same constructor, state_dict keys and `decode()` contract as the reference's `StructureTokenDecoder`.  The torch
sub-modules are parameter containers for `load_state_dict`; the arithmetic runs in libesmk.so:

    esmk_embed -> esmk_trunk_forward (30 blocks, d 1280, no residue scaling)
      -> esmk_head_forward (ffn1/GELU/norm/proj) -> esmk_dim6_frames            tensor7_affine, bb_pred
      -> esmk_head_forward (dense/GELU/norm/output) -> esmk_categorical_mean    plddt
      -> esmk_layernorm(bf16) -> esmk_gemm (downproject) -> esmk_pairwise_pae   predicted_aligned_error, ptm

The reference's pairwise head materialises three [B, L, L, *] fp32 tensors; the fused kernel writes only
pae [B, L, L] and ptm [B].
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from .engine import DeviceBatch, HeadEngine, _f32, _stream, pack_weight
from .packing import pack_sequence_id
from .transformer_stack import TransformerStack, _Params

VQVAE_CODEBOOK_SIZE = 4096  # esm/utils/constants/esm3.py:13-24
VQVAE_SPECIAL_TOKENS = {"MASK": 4096, "EOS": 4097, "BOS": 4098, "PAD": 4099, "CHAINBREAK": 4100}
VQVAE_DIRECTION_LOSS_BINS = 16
VQVAE_PAE_BINS = 64
VQVAE_MAX_PAE_BIN = 31.0
VQVAE_PLDDT_BINS = 50


class _Dim6RotStructureHead(_Params):  # structure_proj.py:13-28
    def __init__(self, d):
        super().__init__()
        self.ffn1 = nn.Linear(d, d)
        self.norm = nn.LayerNorm(d)
        self.proj = nn.Linear(d, 9 + 7 * 2)


class _PairwisePredictionHead(_Params):  # vqvae.py:52-69
    def __init__(self, d, downproject_dim, hidden_dim, n_bins):
        super().__init__()
        self.downproject = nn.Linear(d, downproject_dim, bias=False)
        self.linear1 = nn.Linear(downproject_dim, hidden_dim, bias=False)
        self.norm = nn.LayerNorm(hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, n_bins, bias=False)


class _RegressionHead(_Params):  # vqvae.py:98-104
    def __init__(self, d, out):
        super().__init__()
        self.dense = nn.Linear(d, d)
        self.norm = nn.LayerNorm(d)
        self.output = nn.Linear(d, out)


class StructureTokenDecoder(nn.Module):
    def __init__(self, d_model: int, n_heads: int, n_layers: int):
        super().__init__()
        self.decoder_channels = d_model
        self.vqvae_codebook_size = VQVAE_CODEBOOK_SIZE
        self.special_tokens = VQVAE_SPECIAL_TOKENS
        self.max_pae_bin = VQVAE_MAX_PAE_BIN
        self.embed = nn.Embedding(self.vqvae_codebook_size + len(self.special_tokens), d_model)
        self.decoder_stack = TransformerStack(d_model, n_heads, 1, n_layers, scale_residue=False, n_layers_geom=0)
        self.affine_output_projection = _Dim6RotStructureHead(d_model)
        self.pairwise_bins = [64, VQVAE_DIRECTION_LOSS_BINS * 6, VQVAE_PAE_BINS]
        self.pairwise_classification_head = _PairwisePredictionHead(d_model, 128, 128, sum(self.pairwise_bins))
        self.plddt_head = _RegressionHead(d_model, VQVAE_PLDDT_BINS)
        self.trans_scale_factor = 10.0  # vqvae.py:341-343
        self._eng = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._eng = None

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.embed.weight.device

    def _engines(self):
        if self._eng is None:
            dev = self.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the model to cuda (there is no CPU path)")
            sd = dict(self.state_dict())
            d = self.decoder_channels
            pw = "pairwise_classification_head."
            lo = sum(self.pairwise_bins[:2])  # the pae bins are the last split of linear2's output (vqvae.py:407-410)
            w2 = sd[pw + "linear2.weight"].detach().to(dev, torch.float32)[lo:lo + VQVAE_PAE_BINS]
            g, b = sd[pw + "norm.weight"].detach().to(dev, torch.float32), sd[pw + "norm.bias"].detach().to(dev, torch.float32)
            self._eng = dict(
                embed=_f32(sd["embed.weight"], dev),
                frames=HeadEngine(sd, "affine_output_projection.", d, 23, dev, names=("ffn1", "norm", "proj")),
                plddt=HeadEngine(sd, "plddt_head.", d, VQVAE_PLDDT_BINS, dev, names=("dense", "norm", "output")),
                down=pack_weight(sd[pw + "downproject.weight"], dev),
                w1=sd[pw + "linear1.weight"].detach().to(dev, torch.bfloat16).contiguous(),
                # LayerNorm scale / shift of the pair MLP folded into its output layer (include/esmk.h)
                w2=(w2 * g[None, :]).to(torch.bfloat16).contiguous(),
                b2=(w2 @ b).contiguous(),
            )
        return self.decoder_stack.engine(), self._eng

    @torch.no_grad()
    def decode(self, structure_tokens: torch.Tensor, attention_mask: torch.Tensor | None = None,
               sequence_id: torch.Tensor | None = None):
        """structure_tokens int64 [B, L] (BOS ... EOS, PAD after) -> dict(tensor7_affine [B, L, 12],
        bb_pred [B, L, 3, 3], plddt [B, L], ptm [B], predicted_aligned_error [B, L, L]), all fp32."""
        assert structure_tokens.ndim == 2
        B, Ln = structure_tokens.shape
        dev = structure_tokens.device
        if attention_mask is None:
            attention_mask = torch.ones_like(structure_tokens, dtype=torch.bool)
        attention_mask = attention_mask.bool()
        st = self.special_tokens
        last = structure_tokens[torch.arange(B, device=dev), attention_mask.sum(1) - 1]
        ok = torch.stack([structure_tokens[:, 0].eq(st["BOS"]).all(), last.eq(st["EOS"]).all(),
                          (structure_tokens >= 0).all(), (structure_tokens < self.embed.num_embeddings).all()]).tolist()
        assert ok[0], "First token in structure_tokens must be BOS token"
        assert ok[1], "Last token in structure_tokens must be EOS token"
        assert ok[2], ("All structure tokens set to -1 should be replaced with BOS, EOS, PAD, or MASK tokens by now, "
                       "but that isn't the case!")
        if not ok[3]:
            raise IndexError("structure_tokens out of range for the embedding table")
        trunk, e = self._engines()
        lib, d, M = L.lib(), self.decoder_channels, B * Ln
        # the reference's mask is sequence_id equality (zeros by default: pads attend and are attended, vqvae.py:396)
        sid_np = None if sequence_id is None else sequence_id.detach().cpu().numpy()
        pb = pack_sequence_id(sid_np, B, Ln, drop_false=False)
        assert pb.gather is None
        batch = DeviceBatch(pb, dev)
        tok = structure_tokens.reshape(-1).to(torch.int64).contiguous()
        x = torch.empty((M, d), dtype=torch.float32, device=dev)
        L.check(lib.esmk_embed(L.ptr(tok), L.ptr(e["embed"]), self.embed.num_embeddings, L.ptr(x), M, d, _stream()), "esmk_embed")
        post, pre, _ = trunk.forward(x, batch, want_post=True, want_hiddens=False)
        # backbone frames
        proj = e["frames"].forward(post)
        affine = torch.empty((M, 12), dtype=torch.float32, device=dev)
        bb = torch.empty((M, 3, 3), dtype=torch.float32, device=dev)
        L.check(lib.esmk_dim6_frames(L.ptr(proj), proj.shape[1], self.trans_scale_factor, L.ptr(affine), L.ptr(bb), M,
                                     _stream()), "esmk_dim6_frames")
        # pLDDT
        pl_logits = e["plddt"].forward(post)
        plddt = torch.empty(M, dtype=torch.float32, device=dev)
        L.check(lib.esmk_categorical_mean(L.ptr(pl_logits), M, VQVAE_PLDDT_BINS, 0.0, 1.0, L.ptr(plddt), _stream()),
                "esmk_categorical_mean")
        # pairwise head -> predicted aligned error, pTM
        xb = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
        L.check(lib.esmk_layernorm(L.ptr(pre), L.F32, L.ptr(trunk._norm), None, L.ptr(xb), L.BF16, M, d, 1e-5, _stream()),
                "esmk_layernorm")
        qk = torch.empty((M, 128), dtype=torch.float32, device=dev)
        L.check(lib.esmk_gemm(L.ptr(xb), d, L.ptr(e["down"]), M, 128, d, L.EPI_F32, None, 1.0, L.ptr(qk), _stream()), "esmk_gemm")
        aa = (structure_tokens < min(st.values())).to(torch.uint8).contiguous()
        pae = torch.empty((B, Ln, Ln), dtype=torch.float32, device=dev)
        ptm = torch.empty(B, dtype=torch.float32, device=dev)
        L.check(lib.esmk_pairwise_pae(L.ptr(qk), L.ptr(e["w1"]), L.ptr(e["w2"]), L.ptr(e["b2"]), L.ptr(aa), B, Ln,
                                      self.max_pae_bin, L.ptr(pae), L.ptr(ptm), _stream()), "esmk_pairwise_pae")
        return dict(tensor7_affine=affine.view(B, Ln, 12), bb_pred=bb.view(B, Ln, 3, 3), plddt=plddt.view(B, Ln), ptm=ptm,
                    predicted_aligned_error=pae)


def decode_structure(structure_tokens: torch.Tensor, structure_decoder: StructureTokenDecoder):
    """esm/utils/decoding.py:138-169 up to the backbone: one protein's structure tokens [L] (BOS ... EOS) ->
    (backbone N/CA/C coordinates [L-2, 3, 3], plddt [L-2], ptm, pae [1, L, L]).  Building the all-atom chain
    (`ProteinChain.infer_oxygen`) belongs to the reference's structure utilities and is out of scope here."""
    if structure_tokens.ndim != 1:
        raise ValueError(f"Only one structure can be decoded at a time, got structure tokens of shape {structure_tokens.size()}")
    out = structure_decoder.decode(structure_tokens.unsqueeze(0))
    return out["bb_pred"][0, 1:-1], out["plddt"][0, 1:-1], out["ptm"], out["predicted_aligned_error"]


# ------------------------------------------------------------------------------------------------------------
class _GeomBias(_Params):  # GeometricReasoningOriginalImpl(bias=True), geom_attention.py:38-50
    def __init__(self, d, v_heads):
        super().__init__()
        self.s_norm = nn.LayerNorm(d)
        self.proj = nn.Linear(d, 15 * v_heads)
        self.out_proj = nn.Linear(3 * v_heads, d)
        self.distance_scale_per_head = nn.Parameter(torch.zeros(v_heads))
        self.rotation_scale_per_head = nn.Parameter(torch.zeros(v_heads))


class _GeomBlock(_Params):  # UnifiedTransformerBlock(use_geom_attn, no plain attn, bias=True, expansion 4), vqvae.py:147-155
    def __init__(self, d, h, v_heads):
        super().__init__()
        self.geom_attn = _GeomBias(d, v_heads)
        self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 2 * h), nn.Identity(), nn.Linear(h, d))


class _GeometricEncoderStack(_Params):
    def __init__(self, d, h, v_heads, n_layers):
        super().__init__()
        self.blocks = nn.ModuleList([_GeomBlock(d, h, v_heads) for _ in range(n_layers)])


class _Codebook(_Params):  # EMACodebook buffers, codebook.py:18-20
    def __init__(self, n_codes, dim):
        super().__init__()
        self.register_buffer("embeddings", torch.randn(n_codes, dim))
        self.register_buffer("N", torch.zeros(n_codes))
        self.register_buffer("z_avg", self.embeddings.data.clone())


class _RelPos(_Params):  # RelativePositionEmbedding, vqvae.py:24-29
    def __init__(self, bins, dim):
        super().__init__()
        self.bins = bins
        self.embedding = nn.Embedding(2 * bins + 2, dim)


class StructureTokenEncoder(nn.Module):
    """Backbone coordinates -> structure tokens.  Same constructor, state_dict keys and `encode()` contract as the
    reference (vqvae.py:179-322).  Per residue: its 16 nearest residues (`esmk_knn_neighbourhoods`), relative-position
    embeddings (`esmk_embed`), `n_layers` geometric blocks over the [B*L, 16] neighbourhoods - LayerNorm, projection,
    geometric attention with mask_and_zero_frameless=False (`esmk_geom_attn_ex`), output projection, SwiGLU FFN, every
    Linear with its bias in the GEMM epilogue - then the query node's row goes through pre_vq_proj and the nearest of
    the 4096 codes is taken in fp32 (`esmk_vq_quantize`)."""

    MAX_RESIDUES_PER_CALL = 16384  # neighbourhood tokens per pass = 16 x this (scratch ~ 4 GB at 1024 channels)

    def __init__(self, d_model: int, n_heads: int, v_heads: int, n_layers: int, d_out: int, n_codes: int):
        super().__init__()
        self.d_model, self.v_heads, self.n_layers, self.d_out, self.n_codes = d_model, v_heads, n_layers, d_out, n_codes
        self.ffn_hidden = int(((4 * d_model) + 255) // 256 * 256)  # swiglu_correction_fn(4, d), blocks.py:10-12
        self.transformer = _GeometricEncoderStack(d_model, self.ffn_hidden, v_heads, n_layers)
        self.pre_vq_proj = nn.Linear(d_model, d_out)
        self.codebook = _Codebook(n_codes, d_out)
        self.relative_positional_embedding = _RelPos(32, d_model)
        self.knn = 16
        self._eng = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._eng = None

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.pre_vq_proj.weight.device

    def _engine(self):
        if self._eng is None:
            dev = self.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the model to cuda (there is no CPU path)")
            sd = dict(self.state_dict())
            layers = []
            for i in range(self.n_layers):
                b = f"transformer.blocks.{i}."
                layers.append(dict(
                    sn_g=_f32(sd[b + "geom_attn.s_norm.weight"], dev), sn_b=_f32(sd[b + "geom_attn.s_norm.bias"], dev),
                    w_proj=pack_weight(sd[b + "geom_attn.proj.weight"], dev), b_proj=_f32(sd[b + "geom_attn.proj.bias"], dev),
                    w_out=pack_weight(sd[b + "geom_attn.out_proj.weight"], dev), b_out=_f32(sd[b + "geom_attn.out_proj.bias"], dev),
                    dist=_f32(sd[b + "geom_attn.distance_scale_per_head"], dev),
                    rot=_f32(sd[b + "geom_attn.rotation_scale_per_head"], dev),
                    ln_g=_f32(sd[b + "ffn.0.weight"], dev), ln_b=_f32(sd[b + "ffn.0.bias"], dev),
                    w1=pack_weight(sd[b + "ffn.1.weight"], dev, L.PACK_SWIGLU), b1=_f32(sd[b + "ffn.1.bias"], dev),
                    w2=pack_weight(sd[b + "ffn.3.weight"], dev), b2=_f32(sd[b + "ffn.3.bias"], dev)))
            codes = _f32(sd["codebook.embeddings"], dev)
            self._eng = dict(layers=layers, rel=_f32(sd["relative_positional_embedding.embedding.weight"], dev),
                             w_t=_f32(sd["pre_vq_proj.weight"], dev).t().contiguous(), b=_f32(sd["pre_vq_proj.bias"], dev),
                             codes_t=codes.t().contiguous(), code_sq=(codes ** 2).sum(-1).contiguous())
        return self._eng

    # ---- kNN graph (reference: a static method returning (edges, edge_mask)) -------------------------------
    def _neighbourhoods(self, coords, affine, affine_mask, padding_mask, sequence_id, residue_index, E):
        B, Ln = affine_mask.shape
        dev = coords.device
        n = B * Ln * E
        out = dict(edges=torch.empty((B, Ln, E), dtype=torch.int64, device=dev),
                   edge_mask=torch.empty((B, Ln, E), dtype=torch.uint8, device=dev),
                   affine=torch.empty((n, 12), dtype=torch.float32, device=dev),
                   mask=torch.empty(n, dtype=torch.uint8, device=dev), seq=torch.empty(n, dtype=torch.int32, device=dev),
                   rel=torch.empty(n, dtype=torch.int64, device=dev))
        sid = None if sequence_id is None else sequence_id.to(torch.int32).contiguous()
        res = None if residue_index is None else residue_index.to(torch.int64).contiguous()
        L.check(L.lib().esmk_knn_neighbourhoods(
            L.ptr(coords), L.ptr(affine_mask), L.ptr(padding_mask), L.ptr(sid), L.ptr(res), L.ptr(affine), B, Ln, E,
            self.relative_positional_embedding.bins, L.ptr(out["edges"]), L.ptr(out["edge_mask"]), L.ptr(out["affine"]),
            L.ptr(out["mask"]), L.ptr(out["seq"]), L.ptr(out["rel"]), _stream()), "esmk_knn_neighbourhoods")
        return out

    @torch.no_grad()
    def find_knn_edges(self, coords, padding_mask, coord_mask, sequence_id=None, knn: int | None = None):
        """vqvae.py:265-289 -> (edges int64 [B, L, min(knn, L)], edge_mask bool)."""
        assert knn is not None, "Must specify a non-null knn to find_knn_edges"
        B, Ln = coord_mask.shape
        c = coords[..., :3, :].to(torch.float32).contiguous()
        aff = torch.zeros((B * Ln, 12), dtype=torch.float32, device=c.device)
        nb = self._neighbourhoods(c, aff, coord_mask.to(torch.uint8).contiguous(), padding_mask.to(torch.uint8).contiguous(),
                                  sequence_id, None, min(knn, Ln))
        return nb["edges"], nb["edge_mask"].bool()

    @torch.no_grad()
    def encode(self, coords: torch.Tensor, attention_mask: torch.Tensor | None = None,
               sequence_id: torch.Tensor | None = None, residue_index: torch.Tensor | None = None,
               _z_out: list | None = None):
        """coords f32 [B, L, >=3, 3] (N, CA, C first; NaN / inf where unknown) -> (z_q f32 [B, L, d_out],
        min_encoding_indices int64 [B, L]).  _z_out (not in the reference): a list that receives the pre-quantisation
        vectors z f32 [B, L, d_out] (tests)."""
        assert coords.ndim == 4 and coords.size(-1) == 3 and coords.size(-2) >= 3, "need N, CA, C"
        B, Ln = coords.shape[:2]
        dev = coords.device
        e, lib, d, H = self._engine(), L.lib(), self.d_model, self.v_heads
        c = coords[..., :3, :].to(torch.float32).contiguous()
        affine = torch.empty((B * Ln, 12), dtype=torch.float32, device=dev)
        amask = torch.empty((B, Ln), dtype=torch.uint8, device=dev)
        L.check(lib.esmk_frames(L.ptr(c), L.ptr(affine), L.ptr(amask), B, Ln, _stream()), "esmk_frames")
        pad = torch.zeros((B, Ln), dtype=torch.uint8, device=dev) if attention_mask is None \
            else (~attention_mask.bool()).to(torch.uint8).contiguous()
        E = min(self.knn, Ln)
        nb = self._neighbourhoods(c, affine, amask, pad, sequence_id, residue_index, E)
        zq = torch.empty((B * Ln, self.d_out), dtype=torch.float32, device=dev)
        idx = torch.empty(B * Ln, dtype=torch.int64, device=dev)
        z = torch.empty((B * Ln, self.d_out), dtype=torch.float32, device=dev) if _z_out is not None else None
        ldg = (3 * H + 63) // 64 * 64
        step = self.MAX_RESIDUES_PER_CALL
        for r0 in range(0, B * Ln, step):
            R = min(step, B * Ln - r0)
            M = R * E
            sl = slice(r0 * E, r0 * E + M)
            x = torch.empty((M, d), dtype=torch.float32, device=dev)
            L.check(lib.esmk_embed(L.ptr(nb["rel"][sl]), L.ptr(e["rel"]), e["rel"].shape[0], L.ptr(x), M, d, _stream()), "esmk_embed")
            k_aff, k_mask, k_seq = nb["affine"][sl], nb["mask"][sl], nb["seq"][sl]
            chain = torch.zeros(M, dtype=torch.int32, device=dev)  # vqvae.py:237-239
            row_cu = torch.arange(0, M + 1, E, dtype=torch.int32, device=dev)
            xb = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
            proj = torch.empty((M, 15 * H), dtype=torch.float32, device=dev)
            go = torch.empty((M, ldg), dtype=torch.bfloat16, device=dev)
            hb = torch.empty((M, self.ffn_hidden), dtype=torch.bfloat16, device=dev)
            s = _stream()
            for ly in e["layers"]:
                L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(ly["sn_g"]), L.ptr(ly["sn_b"]), L.ptr(xb), L.BF16, M, d, 1e-5, s), "esmk_layernorm")
                L.check(lib.esmk_gemm(L.ptr(xb), d, L.ptr(ly["w_proj"]), M, 15 * H, d, L.EPI_F32, L.ptr(ly["b_proj"]), 1.0, L.ptr(proj), s), "esmk_gemm")
                L.check(lib.esmk_geom_attn_ex(L.ptr(proj), L.ptr(k_aff), L.ptr(k_mask), L.ptr(k_seq), L.ptr(chain), L.ptr(row_cu),
                                              R, E, L.ptr(ly["dist"]), L.ptr(ly["rot"]), L.ptr(go), ldg, None, M, H, 0, s),
                        "esmk_geom_attn_ex")
                L.check(lib.esmk_gemm(L.ptr(go), ldg, L.ptr(ly["w_out"]), M, d, 3 * H, L.EPI_RESIDUAL, L.ptr(ly["b_out"]), 1.0, L.ptr(x), s), "esmk_gemm")
                L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(ly["ln_g"]), L.ptr(ly["ln_b"]), L.ptr(xb), L.BF16, M, d, 1e-5, s), "esmk_layernorm")
                L.check(lib.esmk_gemm(L.ptr(xb), d, L.ptr(ly["w1"]), M, 2 * self.ffn_hidden, d, L.EPI_SWIGLU, L.ptr(ly["b1"]), 1.0, L.ptr(hb), s), "esmk_gemm")
                L.check(lib.esmk_gemm(L.ptr(hb), self.ffn_hidden, L.ptr(ly["w2"]), M, d, self.ffn_hidden, L.EPI_RESIDUAL, L.ptr(ly["b2"]), 1.0, L.ptr(x), s), "esmk_gemm")
            L.check(lib.esmk_vq_quantize(L.ptr(x), E * d, L.ptr(amask.view(-1)[r0:r0 + R]), L.ptr(e["w_t"]), L.ptr(e["b"]),
                                         L.ptr(e["codes_t"]), L.ptr(e["code_sq"]), R, d, self.d_out, self.n_codes,
                                         None if z is None else L.ptr(z[r0:r0 + R]),
                                         L.ptr(zq[r0:r0 + R]), L.ptr(idx[r0:r0 + R]), s), "esmk_vq_quantize")
        if _z_out is not None:
            _z_out.append(z.view(B, Ln, self.d_out))
        return zq.view(B, Ln, self.d_out), idx.view(B, Ln)
