"""Function-token decoder forward (reference esm/models/function_decoder.py:21-176).

`FunctionTokenDecoder.forward(token_ids [B, depth]) -> {keyword_logits, keyword_tfidf, interpro_logits}` with the
reference's state_dict keys.  Its stack is the block variant `bias=True, qk_layernorm=False, ffn_type="gelu"`,
8 heads of dim 128 over sequences of 64 single-bit tokens, so every row of the input (one residue's 8 function
tokens) is an independent 64-token sequence:

    bits -> esmk_embed -> per block [ esmk_layernorm -> esmk_gemm(+bias) -> esmk_attn_l64_d128 -> esmk_gemm(+bias, +=x)
                                      -> esmk_layernorm -> esmk_gemm(bias+GELU) -> esmk_gemm(+bias, +=x) ]
         -> esmk_layernorm (final norm) -> esmk_segment_mean (mean over the 64 tokens) -> 3 x esmk_head_forward_ex

`decode()` (logits -> InterPro / keyword annotations) needs the InterPro entry list and keyword vocabulary files,
which are downloaded data, not code: it is not built (the reference's constructor reads them too; ours does not).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _lib as L
from .engine import _f32, _stream, pack_weight, rotary_tables, segment_mean
from .transformer_stack import _Params


@dataclass(frozen=True)
class FunctionTokenDecoderConfig:  # function_decoder.py:21-51 minus the two data-file paths
    d_model: int = 1024
    n_heads: int = 8
    n_layers: int = 3
    function_token_vocab_size: int = 260
    function_token_depth: int = 8
    num_interpro_classes: int = 29026
    keyword_vocabulary_size: int = 58641
    unpack_lsh_bits: bool = True
    num_special_tokens: int = 4
    bits_per_token: int = 8


class _AttnBias(_Params):  # MultiHeadAttention(bias=True, qk_layernorm=False), attention.py:17-38
    def __init__(self, d):
        super().__init__()
        self.layernorm_qkv = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 3 * d))
        self.out_proj = nn.Linear(d, d)


class _GeluBlock(_Params):  # UnifiedTransformerBlock(bias=True, qk_layernorm=False, ffn_type="gelu"), blocks.py:41-48
    def __init__(self, d, hidden):
        super().__init__()
        self.attn = _AttnBias(d)
        self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, hidden), nn.Identity(), nn.Linear(hidden, d))


class _GeluStack(_Params):
    def __init__(self, d, hidden, n_layers):
        super().__init__()
        self.blocks = nn.ModuleList([_GeluBlock(d, hidden) for _ in range(n_layers)])
        self.norm = nn.LayerNorm(d, bias=False)


def _head_params(d, out, hidden):  # RegressionHead(d_model, output_dim, hidden_dim), regression_head.py:4-22
    return nn.Sequential(nn.Linear(d, hidden), nn.Identity(), nn.LayerNorm(hidden), nn.Linear(hidden, out))


class FunctionTokenDecoder(nn.Module):
    def __init__(self, config: FunctionTokenDecoderConfig | None = None):
        super().__init__()
        self.config = config = config or FunctionTokenDecoderConfig()
        d = config.d_model
        if not config.unpack_lsh_bits or config.function_token_depth * config.bits_per_token != 64:
            raise NotImplementedError("built for the released configuration: LSH bits unpacked into 64 single-bit tokens")
        if d != 128 * config.n_heads:
            raise NotImplementedError("the decoder's attention kernel is built for head dim 128 (d_model = 128 * n_heads)")
        self.embedding = nn.Embedding(2 * config.function_token_depth * config.bits_per_token, d)
        self.ffn_hidden = int(4 * d)  # gelu_ln_ffn(expansion_ratio=4), blocks.py:41-42
        self.decoder = _GeluStack(d, self.ffn_hidden, config.n_layers)
        self.heads = nn.ModuleDict({
            "keyword_logits": _head_params(d, config.keyword_vocabulary_size, 4 * d),
            "keyword_tfidf": _head_params(d, config.keyword_vocabulary_size, 4 * d),
            "interpro_logits": _head_params(d, config.num_interpro_classes, 4 * d),
        })
        self._eng = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._eng = None

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self) -> torch.device:
        return self.embedding.weight.device

    def _engine(self):
        if self._eng is None:
            dev = self.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the model to cuda (there is no CPU path)")
            sd = dict(self.state_dict())
            layers = []
            for i in range(self.config.n_layers):
                b = f"decoder.blocks.{i}."
                layers.append(dict(
                    ln1_g=_f32(sd[b + "attn.layernorm_qkv.0.weight"], dev), ln1_b=_f32(sd[b + "attn.layernorm_qkv.0.bias"], dev),
                    w_qkv=pack_weight(sd[b + "attn.layernorm_qkv.1.weight"], dev), b_qkv=_f32(sd[b + "attn.layernorm_qkv.1.bias"], dev),
                    w_o=pack_weight(sd[b + "attn.out_proj.weight"], dev), b_o=_f32(sd[b + "attn.out_proj.bias"], dev),
                    ln2_g=_f32(sd[b + "ffn.0.weight"], dev), ln2_b=_f32(sd[b + "ffn.0.bias"], dev),
                    w1=pack_weight(sd[b + "ffn.1.weight"], dev), b1=_f32(sd[b + "ffn.1.bias"], dev),
                    w2=pack_weight(sd[b + "ffn.3.weight"], dev), b2=_f32(sd[b + "ffn.3.bias"], dev)))
            heads = {}
            for name in self.heads:
                p = f"heads.{name}."
                keep = [pack_weight(sd[p + "0.weight"], dev), _f32(sd[p + "0.bias"], dev), _f32(sd[p + "2.weight"], dev),
                        _f32(sd[p + "2.bias"], dev), pack_weight(sd[p + "3.weight"], dev), _f32(sd[p + "3.bias"], dev)]
                h = L.EsmkHead()
                h.w0, h.b0, h.ln_g, h.ln_b, h.w3, h.b3 = [L.ptr(t) for t in keep]
                h.out_dim = sd[p + "3.weight"].shape[0]
                heads[name] = (h, keep)
            cos, sin = rotary_tables(64, dim=128)
            self._eng = dict(layers=layers, heads=heads, embed=_f32(sd["embedding.weight"], dev),
                             norm=_f32(sd["decoder.norm.weight"], dev), cos=cos.to(dev), sin=sin.to(dev))
        return self._eng

    def _inputs(self, token_ids: torch.Tensor) -> torch.Tensor:
        """function_decoder.py:141-163: every LSH byte becomes 8 single-bit tokens with their own vocabulary slots."""
        cfg = self.config
        lsh = token_ids - cfg.num_special_tokens
        bits = torch.cat([torch.bitwise_and(lsh, 1 << hob).gt(0).to(torch.int64) for hob in range(cfg.bits_per_token)], dim=1)
        inputs = 2 * torch.arange(bits.shape[1], device=token_ids.device)[None, :] + bits
        where_special = (token_ids < cfg.num_special_tokens).any(dim=1, keepdim=True)
        return torch.where(where_special, torch.zeros_like(inputs), inputs)

    @torch.no_grad()
    def forward(self, token_ids: torch.Tensor) -> dict[str, torch.Tensor]:
        """token_ids int64 [B, function_token_depth] -> fp32 logits [B, *] per head."""
        assert token_ids.ndim == 2 and token_ids.shape[1] == self.config.function_token_depth
        e, lib, d, hid = self._engine(), L.lib(), self.config.d_model, self.ffn_hidden
        dev = token_ids.device
        B = token_ids.shape[0]
        M = B * 64
        tok = self._inputs(token_ids).reshape(-1).contiguous()
        s = _stream()
        x = torch.empty((M, d), dtype=torch.float32, device=dev)
        L.check(lib.esmk_embed(L.ptr(tok), L.ptr(e["embed"]), e["embed"].shape[0], L.ptr(x), M, d, s), "esmk_embed")
        xb = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
        qkv = torch.empty((M, 3 * d), dtype=torch.bfloat16, device=dev)
        ctx = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
        hb = torch.empty((M, hid), dtype=torch.bfloat16, device=dev)
        for ly in e["layers"]:
            L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(ly["ln1_g"]), L.ptr(ly["ln1_b"]), L.ptr(xb), L.BF16, M, d, 1e-5, s), "esmk_layernorm")
            L.check(lib.esmk_gemm(L.ptr(xb), d, L.ptr(ly["w_qkv"]), M, 3 * d, d, L.EPI_BF16, L.ptr(ly["b_qkv"]), 1.0, L.ptr(qkv), s), "esmk_gemm")
            L.check(lib.esmk_attn_l64_d128(L.ptr(qkv), L.ptr(ctx), L.ptr(e["cos"]), L.ptr(e["sin"]), B, d, self.config.n_heads, s),
                    "esmk_attn_l64_d128")
            L.check(lib.esmk_gemm(L.ptr(ctx), d, L.ptr(ly["w_o"]), M, d, d, L.EPI_RESIDUAL, L.ptr(ly["b_o"]), 1.0, L.ptr(x), s), "esmk_gemm")
            L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(ly["ln2_g"]), L.ptr(ly["ln2_b"]), L.ptr(xb), L.BF16, M, d, 1e-5, s), "esmk_layernorm")
            L.check(lib.esmk_gemm(L.ptr(xb), d, L.ptr(ly["w1"]), M, hid, d, L.EPI_BIAS_GELU, L.ptr(ly["b1"]), 1.0, L.ptr(hb), s), "esmk_gemm")
            L.check(lib.esmk_gemm(L.ptr(hb), hid, L.ptr(ly["w2"]), M, d, hid, L.EPI_RESIDUAL, L.ptr(ly["b2"]), 1.0, L.ptr(x), s), "esmk_gemm")
        post = torch.empty_like(x)
        L.check(lib.esmk_layernorm(L.ptr(x), L.F32, L.ptr(e["norm"]), None, L.ptr(post), L.F32, M, d, 1e-5, s), "esmk_layernorm")
        pooled = segment_mean(post, torch.arange(0, M + 1, 64, dtype=torch.int32, device=dev))  # torch.mean(encoding, dim=1)
        out = {}
        for name, (h, _keep) in e["heads"].items():
            logits = torch.empty((B, h.out_dim), dtype=torch.float32, device=dev)
            need = lib.esmk_head_ex_workspace_bytes(B, d, 4 * d, h.out_dim)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            L.check(lib.esmk_head_forward_ex(C.byref(h), L.ptr(pooled), L.ptr(logits), B, d, 4 * d, L.ptr(ws), need, s),
                    "esmk_head_forward_ex")
            out[name] = logits
        return out

    def decode(self, *a, **k):
        raise NotImplementedError("FunctionTokenDecoder.decode needs the InterPro entry list and keyword vocabulary data "
                                  "files (function_decoder.py:62-73); only forward() is built")
