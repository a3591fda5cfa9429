"""TransformerStack with the reference's constructor, state_dict keys and forward contract
(reference esm/layers/transformer_stack.py:10-94, blocks.py:51-157, attention.py:15-46,
geom_attention.py:10-50) running on the HIP trunk engine.

The torch sub-modules below are parameter CONTAINERS only - they exist so that `load_state_dict` accepts
the reference's checkpoints unchanged.  Their forward() is never called; the arithmetic is in libesmk.so.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from .affine3d import Affine3D
from .engine import DeviceBatch, TrunkEngine, VariantEngine
from .packing import PackedBatch, pack_sequence_id


def swiglu_correction_fn(expansion_ratio: float, d_model: int) -> int:
    return int(((expansion_ratio * d_model) + 255) // 256 * 256)


class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise L.EsmkError("parameter container: the forward pass runs in the HIP engine")


class _Attn(_Params):  # MultiHeadAttention, attention.py:17-38
    def __init__(self, d, bias=False, qk_layernorm=True):
        super().__init__()
        self.layernorm_qkv = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 3 * d, bias=bias))
        self.out_proj = nn.Linear(d, d, bias=bias)
        self.q_ln = nn.LayerNorm(d, bias=bias) if qk_layernorm else nn.Identity()
        self.k_ln = nn.LayerNorm(d, bias=bias) if qk_layernorm else nn.Identity()


class _Geom(_Params):  # GeometricReasoningOriginalImpl, geom_attention.py:10-50
    def __init__(self, d, v_heads, bias=False):
        super().__init__()
        self.s_norm = nn.LayerNorm(d, bias=bias)
        self.proj = nn.Linear(d, 15 * v_heads, bias=bias)
        self.out_proj = nn.Linear(3 * v_heads, d, bias=bias)
        self.distance_scale_per_head = nn.Parameter(torch.zeros(v_heads))
        self.rotation_scale_per_head = nn.Parameter(torch.zeros(v_heads))


class _Block(_Params):  # UnifiedTransformerBlock, blocks.py:73-115
    def __init__(self, d, v_heads, geom, bias=False, qk_layernorm=True, ffn_type="swiglu", expansion_ratio=8 / 3):
        super().__init__()
        self.attn = _Attn(d, bias, qk_layernorm)
        if geom:
            self.geom_attn = _Geom(d, v_heads, bias)
        if ffn_type == "swiglu":
            h = swiglu_correction_fn(expansion_ratio, d)
            self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 2 * h, bias=bias), nn.Identity(),
                                     nn.Linear(h, d, bias=bias))
        elif ffn_type == "gelu":
            h = int(expansion_ratio * d)
            self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, h, bias=bias), nn.Identity(), nn.Linear(h, d, bias=bias))
        else:
            raise ValueError(f"Unknown ffn_type: {ffn_type}")


class TransformerStack(nn.Module):
    """Every constructor option of the reference (transformer_stack.py:26-62).  The ESM C / ESM3 variant (bias=False,
    qk_layernorm=True, ffn_type="swiglu", geometric blocks with mask_and_zero_frameless=True) runs as ONE C call
    (`esmk_trunk_forward`, fused residual deltas); the other block variants of blocks.py:73-115 run block by block over
    the same kernels (`engine.VariantEngine`), bias=True together with qk_layernorm=True included (q_ln / k_ln then
    carry a bias, attention.py:32-33: `esmk_qk_prep_ex`)."""

    def __init__(self, d_model: int, n_heads: int, v_heads: int | None, n_layers: int, n_layers_geom: int = 1,
                 scale_residue: bool = True, mask_and_zero_frameless: bool = False, bias: bool = False,
                 qk_layernorm: bool = True, ffn_type: str = "swiglu", expansion_ratio: float = 8 / 3,
                 use_flash_attn: bool = False):
        super().__init__()
        if ffn_type not in ("swiglu", "gelu"):
            raise ValueError(f"Unknown ffn_type: {ffn_type}")
        if n_layers_geom > 0 and v_heads is None:
            raise ValueError("v_heads must be specified when use_geom_attn is True")
        self.d_model, self.n_heads, self.v_heads, self.n_layers = d_model, n_heads, v_heads, n_layers
        self.n_layers_geom, self.scale_residue, self.expansion_ratio = n_layers_geom, scale_residue, expansion_ratio
        self.mask_and_zero_frameless, self.bias, self.qk_layernorm, self.ffn_type = \
            mask_and_zero_frameless, bias, qk_layernorm, ffn_type
        # the fused whole-trunk call covers the ESM C / ESM3 variant; everything else goes block by block
        self.fused = (not bias) and qk_layernorm and ffn_type == "swiglu" and (n_layers_geom == 0 or mask_and_zero_frameless)
        self.blocks = nn.ModuleList([_Block(d_model, v_heads, i < n_layers_geom, bias, qk_layernorm, ffn_type,
                                            expansion_ratio) for i in range(n_layers)])
        self.norm = nn.LayerNorm(d_model, bias=False)
        self.small_tokens = 1024  # batches up to this size take the library's small-batch path (0 = never)
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    # -- engine lifetime ---------------------------------------------------------------------------
    def _invalidate(self):
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self) -> TrunkEngine:
        if self._engine is None:
            dev = self.norm.weight.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the module to cuda (there is no CPU path)")
            cls = TrunkEngine if self.fused else VariantEngine
            extra = dict(small_tokens=self.small_tokens) if self.fused else dict(mask_and_zero_frameless=self.mask_and_zero_frameless, bias=self.bias,
                                               qk_layernorm=self.qk_layernorm, ffn_type=self.ffn_type)
            self._engine = cls(dict(self.state_dict()), "", self.d_model, self.n_heads, self.n_layers, dev,
                               v_heads=self.v_heads, n_layers_geom=self.n_layers_geom,
                               scale_residue=self.scale_residue, expansion_ratio=self.expansion_ratio, **extra)
        return self._engine

    # -- packed entry point (ESMC / the data-parallel runner) --------------------------------------
    def forward_packed(self, x: torch.Tensor, batch: DeviceBatch, want_post=True, want_hiddens=False):
        return self.engine().forward(x, batch, want_post, want_hiddens)

    # -- reference signature -----------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, sequence_id: torch.Tensor | None = None, affine: Affine3D | None = None,
                affine_mask: torch.Tensor | None = None, chain_id: torch.Tensor | None = None,
                return_hiddens: bool = True):
        """x [B, L, d] -> (post_norm, pre_norm, hiddens) in fp32 (transformer_stack.py:64-94).

        return_hiddens=False (not in the reference) skips the n_layers device copies; hiddens is then []."""
        *batch_dims, d = x.shape
        if len(batch_dims) != 2:
            raise ValueError("x must be [B, L, d]")
        B, Ln = batch_dims
        dev = x.device
        sid_np = None if sequence_id is None else sequence_id.detach().cpu().numpy()
        pb = pack_sequence_id(sid_np, B, Ln, drop_false=False)
        aff = am = ch = None
        if self.n_layers_geom > 0:
            if affine is None or affine_mask is None:
                raise ValueError("affine and affine_mask are required by the geometric-attention block")
            at = affine.tensor if isinstance(affine, Affine3D) else affine
            aff = at.to(torch.float32).reshape(B * Ln, 12).contiguous()
            am = affine_mask.reshape(B * Ln).to(torch.uint8).contiguous()
            if chain_id is None:
                chain_id = torch.ones((B, Ln), dtype=torch.int64, device=dev)
            ch = chain_id.reshape(B * Ln).to(torch.int32).contiguous()
        xf = x.to(torch.float32).reshape(B * Ln, d)
        gidx = None
        if pb.gather is not None:  # non-contiguous ids: tokens of a row sorted by id (packing.pack_sequence_id)
            gidx = torch.from_numpy(pb.gather).to(dev)
            xf = xf[gidx]
            aff, am, ch = (None if t is None else t[gidx].contiguous() for t in (aff, am, ch))
        else:
            xf = xf.clone()
        batch = DeviceBatch(pb, dev, aff, am, ch)
        post, pre, hid = self.engine().forward(xf, batch, want_post=True, want_hiddens=return_hiddens)

        def back(t):  # packed order -> [B, L, d]
            if gidx is None:
                return t.view(B, Ln, d)
            out = torch.empty_like(t)
            out[gidx] = t
            return out.view(B, Ln, d)

        hiddens = [back(hid[i]) for i in range(self.n_layers)] if return_hiddens else []
        return back(post), back(pre), hiddens
