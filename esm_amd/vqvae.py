"""VQ-VAE structure decoder: ESM3 structure tokens -> backbone coordinates, pLDDT, pTM, predicted aligned error
(reference esm/models/vqvae.py:325-437, esm/layers/structure_proj.py,
esm/utils/structure/predicted_aligned_error.py, caller esm/utils/decoding.py:138-169).

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
