"""Device-side engine: packed weights + one C call per trunk forward.

torch is used for device memory, the current HIP stream and tensor views only.  All arithmetic on the
hot path happens inside libesmk.so (hand-written gfx950 kernels); there is no torch fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .packing import PackedBatch


def _stream():
    return torch.cuda.current_stream().cuda_stream


def rotary_tables(n_pos: int, dim: int = 64, base: float = 10000.0):
    """fp32 cos/sin tables computed exactly as the reference does (esm/layers/rotary.py:136-177)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    freqs = torch.outer(t, inv_freq)
    return torch.cos(freqs).contiguous(), torch.sin(freqs).contiguous()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def pack_weight(w: torch.Tensor, device, mode: int = L.PACK_PLAIN) -> torch.Tensor:
    """fp32 [N, K] -> bf16 MFMA layout (esmk_pack_weight)."""
    lib = L.lib()
    w = _f32(w, device)
    N, K = w.shape
    nbytes = lib.esmk_packed_weight_bytes(N, K)
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=device)
    L.check(lib.esmk_pack_weight(L.ptr(w), L.ptr(out), N, K, mode, _stream()), "esmk_pack_weight")
    return out


class DeviceBatch:
    """PackedBatch uploaded to the device + the esmk_batch descriptor."""

    def __init__(self, pb: PackedBatch, device, affine=None, affine_mask=None, chain_id=None):
        self.pb = pb
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
        self.cu_seqlens = i32(pb.cu_seqlens)
        self.positions = i32(pb.positions)
        self.qblk_map = i32(pb.qblk_map.reshape(-1))
        self.row_cu = i32(pb.row_cu)
        self.seq_id = i32(pb.seq_id)
        self.affine = affine  # f32 [N, 12] on device or None
        self.affine_mask = affine_mask  # uint8 [N]
        self.chain_id = chain_id  # int32 [N]
        d = L.EsmkBatch()
        d.n_tokens = pb.n_tokens
        d.n_segments = pb.n_segments
        d.cu_seqlens = L.ptr(self.cu_seqlens)
        d.positions = L.ptr(self.positions)
        d.qblk_map = L.ptr(self.qblk_map)
        d.n_qblk = int(pb.qblk_map.shape[0])
        d.affine = L.ptr(affine)
        d.affine_mask = L.ptr(affine_mask)
        d.seq_id = L.ptr(self.seq_id)
        d.chain_id = L.ptr(chain_id)
        d.row_cu = L.ptr(self.row_cu)
        d.n_rows = pb.B
        d.max_row_len = pb.max_row_len
        d.pad_tail = 0
        d.max_position = int(pb.max_position)
        self.desc = d

    _padded_static: dict = {}

    @classmethod
    def padded(cls, B: int, Ln: int, cu_seqlens: torch.Tensor, device, positions: torch.Tensor | None = None) -> "DeviceBatch":
        """Batch of a padded [B, L] token matrix whose row lengths are known on the DEVICE only
        (esmk_compact_tokens): all B*L positions stay packed, `cu_seqlens` int32 [2B+1] holds (valid tokens,
        padding) segment pairs and `positions` int32 [B*L] every slot's column; everything else depends on (B, L)
        alone and is cached per device - building it costs no host synchronisation."""
        device = torch.device(device)
        key = (B, Ln, device.type, device.index if device.index is not None else torch.cuda.current_device())
        st = cls._padded_static.get(key)
        if st is None:
            if len(cls._padded_static) >= 16:
                cls._padded_static.pop(next(iter(cls._padded_static)))
            nq = (Ln + L.ATTN_QBLK - 1) // L.ATTN_QBLK
            qmap = np.stack([np.repeat(2 * np.arange(B, dtype=np.int32), nq),
                             np.tile(np.arange(nq, dtype=np.int32) * L.ATTN_QBLK, B)], axis=1)
            row_cu = (np.arange(B + 1, dtype=np.int64) * Ln).astype(np.int32)
            pb = PackedBatch(B, Ln, B * Ln, row_cu, np.tile(np.arange(Ln, dtype=np.int32), B), None, row_cu,
                             np.zeros(B * Ln, dtype=np.int32), qmap)
            st = cls(pb, device)
            cls._padded_static[key] = st
        self = object.__new__(cls)
        self.__dict__.update(st.__dict__)
        self.cu_seqlens = cu_seqlens
        d = L.EsmkBatch()
        C.memmove(C.byref(d), C.byref(st.desc), C.sizeof(d))
        d.cu_seqlens = L.ptr(cu_seqlens)
        if positions is not None:
            self.positions = positions
            d.positions = L.ptr(positions)
        d.n_segments = 2 * B
        d.pad_tail = 1
        self.desc = d
        return self


    _rows_static: dict = {}

    @classmethod
    def rows(cls, B: int, Ln: int, device, affine=None, affine_mask=None, chain_id=None) -> "DeviceBatch":
        """Batch of B full rows of Ln tokens with NO sequence_id (every position of a row attends to the whole row,
        attention.py:73-78) plus this call's geometric-attention inputs.  Everything but those inputs depends on (B, Ln)
        alone and is cached on the device: building it costs no host work and no synchronisation.  chain_id None = one
        chain (the zeros ESM3.forward passes, esm3.py:326)."""
        from .packing import pack_sequence_id
        device = torch.device(device)
        key = (B, Ln, device.type, device.index if device.index is not None else torch.cuda.current_device())
        st = cls._rows_static.get(key)
        if st is None:
            if len(cls._rows_static) >= 16:
                cls._rows_static.pop(next(iter(cls._rows_static)))
            st = cls(pack_sequence_id(None, B, Ln, drop_false=False), device)
            st._zero_chain = torch.zeros(B * Ln, dtype=torch.int32, device=device)
            cls._rows_static[key] = st
        self = object.__new__(cls)
        self.__dict__.update(st.__dict__)
        self.affine, self.affine_mask = affine, affine_mask
        self.chain_id = st._zero_chain if chain_id is None else chain_id
        d = L.EsmkBatch()
        C.memmove(C.byref(d), C.byref(st.desc), C.sizeof(d))
        d.affine, d.affine_mask, d.chain_id = L.ptr(self.affine), L.ptr(self.affine_mask), L.ptr(self.chain_id)
        self.desc = d
        return self


class TrunkEngine:
    """Packed TransformerStack weights (reference key names) + esmk_trunk_forward."""

    def __init__(self, weights: dict, prefix: str, d_model: int, n_heads: int, n_layers: int, device,
                 v_heads: int | None = None, n_layers_geom: int = 0, scale_residue: bool = True,
                 expansion_ratio: float = 8 / 3, rope_len: int = 4096, small_tokens: int = 1024):
        """small_tokens: batches of at most this many tokens take the small-batch path of the library (6 launches per
        block, LayerNorms folded into the projections); its LayerNorm-folded weight copies are built on the first such
        batch, so a model that only ever sees large batches never holds them.  0 = off."""
        if d_model != 64 * n_heads:
            raise L.EsmkError(f"head dim must be 64 (d_model={d_model}, n_heads={n_heads})")
        self.lib = L.lib()
        self._src = (weights, prefix, n_layers_geom)
        self._small_tokens = int(small_tokens) if d_model <= 4096 else 0
        self._folded = False
        self.device = torch.device(device)
        self.d, self.H, self.n_layers = d_model, n_heads, n_layers
        self.v_heads = v_heads or 0
        self.h = int(((expansion_ratio * d_model) + 255) // 256 * 256)
        self._keep = []  # device tensors referenced by raw pointers
        self._layers = (L.EsmkLayer * n_layers)()
        dev = self.device

        def f(name):
            t = _f32(weights[prefix + name], dev)
            self._keep.append(t)
            return L.ptr(t)

        def p(name, mode=L.PACK_PLAIN):
            t = pack_weight(weights[prefix + name], dev, mode)
            self._keep.append(t)
            return L.ptr(t)

        for i in range(n_layers):
            b = f"blocks.{i}."
            ly = self._layers[i]
            ly.ln1_g, ly.ln1_b = f(b + "attn.layernorm_qkv.0.weight"), f(b + "attn.layernorm_qkv.0.bias")
            ly.w_qkv = p(b + "attn.layernorm_qkv.1.weight")
            ly.qln_g, ly.kln_g = f(b + "attn.q_ln.weight"), f(b + "attn.k_ln.weight")
            ly.w_o = p(b + "attn.out_proj.weight")
            ly.ln2_g, ly.ln2_b = f(b + "ffn.0.weight"), f(b + "ffn.0.bias")
            ly.w_ffn1 = p(b + "ffn.1.weight", L.PACK_SWIGLU)
            ly.w_ffn2 = p(b + "ffn.3.weight")
            if i < n_layers_geom:
                ly.geom_ln_g = f(b + "geom_attn.s_norm.weight")
                ly.w_geom_proj = p(b + "geom_attn.proj.weight")
                ly.w_geom_out = p(b + "geom_attn.out_proj.weight")
                ly.geom_dist_scale = f(b + "geom_attn.distance_scale_per_head")
                ly.geom_rot_scale = f(b + "geom_attn.rotation_scale_per_head")
        self._norm = _f32(weights[prefix + "norm.weight"], dev)
        self._desc = L.EsmkTrunk()
        t = self._desc
        t.d_model, t.n_heads, t.n_layers, t.ffn_hidden, t.v_heads = d_model, n_heads, n_layers, self.h, self.v_heads
        t.residual_scale = math.sqrt(n_layers / 36) if scale_residue else 1.0
        t.layers = C.cast(self._layers, C.POINTER(L.EsmkLayer))
        t.final_norm_g = L.ptr(self._norm)
        self._set_rope(rope_len)
        self._ws = None

    def _set_rope(self, n_pos: int):
        cos, sin = rotary_tables(n_pos)
        self._cos, self._sin = cos.to(self.device), sin.to(self.device)
        self._desc.rope_cos, self._desc.rope_sin, self._desc.rope_len = L.ptr(self._cos), L.ptr(self._sin), n_pos

    def _fold_layernorms(self):
        """W' = W . diag(gamma) (packed), u = W' . 1, v = W . beta for the three projections that follow a LayerNorm
        (esmk_fold_layernorm): the operands of esmk_gemm_lnfold on the small-batch path."""
        weights, prefix, n_geom = self._src
        dev = self.device

        def fold(wname, gname, bname, mode=L.PACK_PLAIN):
            w = _f32(weights[prefix + wname], dev)
            g = _f32(weights[prefix + gname], dev)
            b = _f32(weights[prefix + bname], dev) if bname else None
            N, K = w.shape
            wf = torch.empty_like(w)
            u = torch.empty(N, dtype=torch.float32, device=dev)
            v = torch.empty(N, dtype=torch.float32, device=dev) if b is not None else None
            L.check(self.lib.esmk_fold_layernorm(L.ptr(w), L.ptr(g), L.ptr(b), N, K, L.ptr(wf), L.ptr(u), L.ptr(v), _stream()),
                    "esmk_fold_layernorm")
            wp = pack_weight(wf, dev, mode)
            self._keep += [wp, u] + ([v] if v is not None else [])
            return L.ptr(wp), L.ptr(u), L.ptr(v)

        for i in range(self.n_layers):
            b = f"blocks.{i}."
            ly = self._layers[i]
            ly.w_qkv_f, ly.qkv_u, ly.qkv_v = fold(b + "attn.layernorm_qkv.1.weight", b + "attn.layernorm_qkv.0.weight",
                                                  b + "attn.layernorm_qkv.0.bias")
            ly.w_ffn1_f, ly.ffn1_u, ly.ffn1_v = fold(b + "ffn.1.weight", b + "ffn.0.weight", b + "ffn.0.bias", L.PACK_SWIGLU)
            if i < n_geom:
                ly.w_geom_proj_f, ly.geom_u, _ = fold(b + "geom_attn.proj.weight", b + "geom_attn.s_norm.weight", None)
        self._desc.small_tokens = self._small_tokens
        self._folded = True
        self._ws = None  # the workspace layout depends on the path

    def enable_probes(self, on: bool = True):
        """Record HIP events around every layer's FFN up-projection GEMM (the dominant kernel)."""
        if on and not getattr(self, "_probe_ev", None):
            n = self.n_layers
            self._probe_ev = ((L.vp * n)(), (L.vp * n)())
            for arr in self._probe_ev:
                for i in range(n):
                    e = L.vp()
                    L.check(self.lib.esmk_event_create(C.byref(e)), "esmk_event_create")
                    arr[i] = e.value
        if on:
            self._desc.probe_start = C.cast(self._probe_ev[0], C.POINTER(L.vp))
            self._desc.probe_stop = C.cast(self._probe_ev[1], C.POINTER(L.vp))
        else:
            self._desc.probe_start = None
            self._desc.probe_stop = None

    def probe_times_ms(self):
        """Per-layer duration of the probed kernel in the LAST forward (synchronises)."""
        out = []
        for i in range(self.n_layers):
            ms = C.c_float()
            L.check(self.lib.esmk_event_elapsed_ms(self._probe_ev[0][i], self._probe_ev[1][i], C.byref(ms)))
            out.append(ms.value)
        return out

    def workspace(self, n_tokens: int) -> torch.Tensor:
        need = self.lib.esmk_trunk_workspace_bytes(C.byref(self._desc), n_tokens)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, x: torch.Tensor, batch: DeviceBatch, want_post: bool = True, want_hiddens: bool = False,
                hidden_layer: int = -1):
        """x f32 [N, d] (modified in place -> pre-norm output).  Returns (post_norm|None, x, hiddens|None).
        hidden_layer >= 0 with want_hiddens: only that block's output is captured, hiddens is [1, N, d]."""
        N = batch.pb.n_tokens
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape == (N, self.d)
        if batch.pb.max_position >= self._desc.rope_len:
            self._set_rope(max(2 * self._desc.rope_len, batch.pb.max_position + 1))
        post = torch.empty_like(x) if want_post else None
        if want_hiddens and not -1 <= hidden_layer < self.n_layers:
            raise IndexError(f"hidden layer {hidden_layer} outside [0, {self.n_layers})")
        n_hid = self.n_layers if hidden_layer < 0 else 1
        hid = torch.empty((n_hid, N, self.d), dtype=torch.float32, device=x.device) if want_hiddens else None
        if N == 0:
            return post, x, hid
        if 0 < N <= self._small_tokens and not self._folded:
            self._fold_layernorms()
        ws = self.workspace(N)
        rc = self.lib.esmk_trunk_forward_ex(C.byref(self._desc), C.byref(batch.desc), L.ptr(x), L.ptr(post), L.ptr(hid),
                                            hidden_layer, L.ptr(ws), ws.numel(), _stream())
        L.check(rc, "esmk_trunk_forward")
        return post, x, hid


class VariantEngine:
    """The block variants of the reference's UnifiedTransformerBlock that the fused trunk call does not cover
    (blocks.py:73-115): bias=True, qk_layernorm=False, ffn_type="gelu", geometric attention with
    mask_and_zero_frameless=False.  Same kernels, one C call per op instead of one per forward:

        esmk_layernorm -> esmk_gemm(+bias) -> esmk_qk_prep (gammas NULL without QK-LayerNorm) -> esmk_attn_varlen
          -> esmk_gemm(residual: x += (ctx.Wo^T + b) / s)
        [esmk_layernorm -> esmk_gemm(f32) -> esmk_geom_attn_ex -> esmk_gemm(residual)]
        esmk_layernorm -> esmk_gemm(SwiGLU | bias+GELU) -> esmk_gemm(residual)

    Same interface as TrunkEngine.forward."""

    def __init__(self, weights: dict, prefix: str, d_model: int, n_heads: int, n_layers: int, device,
                 v_heads: int | None = None, n_layers_geom: int = 0, scale_residue: bool = True,
                 expansion_ratio: float = 8 / 3, rope_len: int = 4096, mask_and_zero_frameless: bool = False,
                 bias: bool = False, qk_layernorm: bool = True, ffn_type: str = "swiglu"):
        if d_model != 64 * n_heads:
            raise L.EsmkError(f"head dim must be 64 (d_model={d_model}, n_heads={n_heads})")
        self.lib = L.lib()
        self.device = dev = torch.device(device)
        self.d, self.H, self.n_layers, self.Hv = d_model, n_heads, n_layers, v_heads or 0
        self.mzf, self.ffn_type = int(bool(mask_and_zero_frameless)), ffn_type
        self.inv_s = 1.0 / (math.sqrt(n_layers / 36) if scale_residue else 1.0)
        if ffn_type == "swiglu":
            self.h = int(((expansion_ratio * d_model) + 255) // 256 * 256)
        else:
            self.h = int(expansion_ratio * d_model)
            if self.h % 64:
                raise L.EsmkError(f"GELU FFN hidden width {self.h} must be a multiple of 64 (the GEMM's K granule)")
        f = lambda k: _f32(weights[prefix + k], dev)
        ob = lambda k: _f32(weights[prefix + k], dev) if bias else None  # optional bias
        self.layers = []
        for i in range(n_layers):
            b = f"blocks.{i}."
            ly = dict(ln1_g=f(b + "attn.layernorm_qkv.0.weight"), ln1_b=f(b + "attn.layernorm_qkv.0.bias"),
                      w_qkv=pack_weight(weights[prefix + b + "attn.layernorm_qkv.1.weight"], dev),
                      b_qkv=ob(b + "attn.layernorm_qkv.1.bias"),
                      gq=f(b + "attn.q_ln.weight") if qk_layernorm else None,
                      gk=f(b + "attn.k_ln.weight") if qk_layernorm else None,
                      bq=ob(b + "attn.q_ln.bias") if qk_layernorm else None,  # bias=True: biased q_ln / k_ln
                      bk=ob(b + "attn.k_ln.bias") if qk_layernorm else None,
                      w_o=pack_weight(weights[prefix + b + "attn.out_proj.weight"], dev), b_o=ob(b + "attn.out_proj.bias"),
                      ln2_g=f(b + "ffn.0.weight"), ln2_b=f(b + "ffn.0.bias"),
                      w1=pack_weight(weights[prefix + b + "ffn.1.weight"], dev,
                                     L.PACK_SWIGLU if ffn_type == "swiglu" else L.PACK_PLAIN),
                      b1=ob(b + "ffn.1.bias"),
                      w2=pack_weight(weights[prefix + b + "ffn.3.weight"], dev), b2=ob(b + "ffn.3.bias"))
            if ffn_type == "gelu" and ly["b1"] is None:  # the bias+GELU epilogue always takes a bias
                ly["b1"] = torch.zeros(self.h, dtype=torch.float32, device=dev)
            if i < n_layers_geom:
                ly.update(sn_g=f(b + "geom_attn.s_norm.weight"), sn_b=ob(b + "geom_attn.s_norm.bias"),
                          w_proj=pack_weight(weights[prefix + b + "geom_attn.proj.weight"], dev),
                          b_proj=ob(b + "geom_attn.proj.bias"),
                          w_out=pack_weight(weights[prefix + b + "geom_attn.out_proj.weight"], dev),
                          b_out=ob(b + "geom_attn.out_proj.bias"),
                          dist=f(b + "geom_attn.distance_scale_per_head"), rot=f(b + "geom_attn.rotation_scale_per_head"))
            self.layers.append(ly)
        self.norm_g = f("norm.weight")
        self._set_rope(rope_len)

    def _set_rope(self, n_pos: int):
        cos, sin = rotary_tables(n_pos)
        self._cos, self._sin, self.rope_len = cos.to(self.device), sin.to(self.device), n_pos

    def forward(self, x: torch.Tensor, batch: DeviceBatch, want_post: bool = True, want_hiddens: bool = False):
        """x f32 [N, d] (modified in place -> pre-norm output).  Returns (post_norm|None, x, hiddens|None)."""
        N, d, H, Hv, h, lib, dev = batch.pb.n_tokens, self.d, self.H, self.Hv, self.h, self.lib, x.device
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape == (N, d)
        if batch.pb.max_position >= self.rope_len:
            self._set_rope(max(2 * self.rope_len, batch.pb.max_position + 1))
        post = torch.empty_like(x) if want_post else None
        hid = torch.empty((self.n_layers, N, d), dtype=torch.float32, device=dev) if want_hiddens else None
        if N == 0:
            return post, x, hid
        s = _stream()
        bf = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
        xb, qkv, ctx, hb = bf(N, d), bf(N, 3 * d), bf(N, d), bf(N, h)
        q_scale = 0.125 * 1.4426950408889634  # 64^-1/2 * log2(e): the attention kernel exponentiates with exp2
        ln = lambda g, b: L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(g), L.ptr(b), L.ptr(xb), L.BF16, N, d, 1e-5, s),
                                  "esmk_layernorm")
        gemm = lambda a, lda, w, n, k, epi, b, alpha, out: L.check(
            lib.esmk_gemm(L.ptr(a), lda, L.ptr(w), N, n, k, epi, L.ptr(b), alpha, L.ptr(out), s), "esmk_gemm")
        for i, ly in enumerate(self.layers):
            ln(ly["ln1_g"], ly["ln1_b"])
            gemm(xb, d, ly["w_qkv"], 3 * d, d, L.EPI_BF16, ly["b_qkv"], 1.0, qkv)
            L.check(lib.esmk_qk_prep_ex(L.ptr(qkv), L.ptr(ly["gq"]), L.ptr(ly["gk"]), L.ptr(ly["bq"]), L.ptr(ly["bk"]),
                                        L.ptr(batch.positions), L.ptr(self._cos), L.ptr(self._sin), self.rope_len, N, d, H,
                                        q_scale, 1e-5, s), "esmk_qk_prep")
            L.check(lib.esmk_attn_varlen(L.ptr(qkv), L.ptr(ctx), L.ptr(batch.cu_seqlens), L.ptr(batch.qblk_map),
                                         batch.desc.n_qblk, N, d, H, s), "esmk_attn_varlen")
            gemm(ctx, d, ly["w_o"], d, d, L.EPI_RESIDUAL, ly["b_o"], self.inv_s, x)
            if "w_proj" in ly:
                if batch.affine is None or batch.affine_mask is None or batch.chain_id is None:
                    raise ValueError("affine, affine_mask and chain_id are required by the geometric-attention block")
                ldg = (3 * Hv + 63) // 64 * 64
                proj = torch.empty((N, 15 * Hv), dtype=torch.float32, device=dev)
                go = bf(N, ldg)
                scratch = torch.empty(lib.esmk_geom_scratch_bytes(N, Hv), dtype=torch.uint8, device=dev)
                ln(ly["sn_g"], ly["sn_b"])
                gemm(xb, d, ly["w_proj"], 15 * Hv, d, L.EPI_F32, ly["b_proj"], 1.0, proj)
                L.check(lib.esmk_geom_attn_ex(L.ptr(proj), L.ptr(batch.affine), L.ptr(batch.affine_mask), L.ptr(batch.seq_id),
                                              L.ptr(batch.chain_id), L.ptr(batch.row_cu), batch.pb.B, batch.pb.max_row_len,
                                              L.ptr(ly["dist"]), L.ptr(ly["rot"]), L.ptr(go), ldg, L.ptr(scratch), N, Hv,
                                              self.mzf, s), "esmk_geom_attn_ex")
                gemm(go, ldg, ly["w_out"], d, 3 * Hv, L.EPI_RESIDUAL, ly["b_out"], self.inv_s, x)
            ln(ly["ln2_g"], ly["ln2_b"])
            if self.ffn_type == "swiglu":
                gemm(xb, d, ly["w1"], 2 * h, d, L.EPI_SWIGLU, ly["b1"], 1.0, hb)
            else:
                gemm(xb, d, ly["w1"], h, d, L.EPI_BIAS_GELU, ly["b1"], 1.0, hb)
            gemm(hb, h, ly["w2"], d, h, L.EPI_RESIDUAL, ly["b2"], self.inv_s, x)
            if hid is not None:
                hid[i].copy_(x)
        if post is not None:
            L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(self.norm_g), None, L.ptr(post), L.F32, N, d, 1e-5, s),
                    "esmk_layernorm")
        return post, x, hid


def segment_mean(x: torch.Tensor, cu: torch.Tensor) -> torch.Tensor:
    """Mean of x f32 [N, d] over the row ranges cu int32 [S+1] (device) -> [S, d] (esmk_segment_mean)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1 and cu.dtype == torch.int32
    S = cu.numel() - 1
    out = torch.empty((S, x.shape[1]), dtype=torch.float32, device=x.device)
    L.check(L.lib().esmk_segment_mean(L.ptr(x), x.stride(0), L.ptr(cu), S, L.ptr(out), x.shape[1], _stream()),
            "esmk_segment_mean")
    return out


class HeadEngine:
    """RegressionHead (reference esm/layers/regression_head.py:17-22) on esmk_head_forward."""

    def __init__(self, weights: dict, prefix: str, d_model: int, out_dim: int, device, names=("0", "2", "3")):
        """names: sub-module names of (first Linear, LayerNorm, output Linear): ("0", "2", "3") in the reference's
        nn.Sequential heads, ("dense", "norm", "output") / ("ffn1", "norm", "proj") in the VQ-VAE decoder's."""
        self.lib = L.lib()
        self.device = torch.device(device)
        self.d, self.out_dim = d_model, out_dim
        lin0, norm, lin1 = (prefix + n for n in names)
        self._keep = [
            pack_weight(weights[lin0 + ".weight"], device), _f32(weights[lin0 + ".bias"], device),
            _f32(weights[norm + ".weight"], device), _f32(weights[norm + ".bias"], device),
            pack_weight(weights[lin1 + ".weight"], device), _f32(weights[lin1 + ".bias"], device),
        ]
        h = L.EsmkHead()
        h.w0, h.b0, h.ln_g, h.ln_b, h.w3, h.b3 = [L.ptr(t) for t in self._keep]
        h.out_dim = out_dim
        self._desc = h
        self._ws = None

    def forward(self, x_norm: torch.Tensor) -> torch.Tensor:
        N = x_norm.shape[0]
        assert x_norm.is_cuda and x_norm.dtype == torch.float32 and x_norm.is_contiguous()
        out = torch.empty((N, self.out_dim), dtype=torch.float32, device=x_norm.device)
        if N == 0:
            return out
        need = self.lib.esmk_head_workspace_bytes(N, self.d)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        rc = self.lib.esmk_head_forward(C.byref(self._desc), L.ptr(x_norm), L.ptr(out), N, self.d, L.ptr(self._ws),
                                        self._ws.numel(), _stream())
        L.check(rc, "esmk_head_forward")
        return out
