"""CPU fp32 restatement of FunctionTokenDecoder.forward (reference
esm/models/function_decoder.py:127-176).  TEST INFRASTRUCTURE ONLY, pinned by oracle/make_golden.py fdec_case
(tests/golden/fdec_*.npz)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .trunk_ref import gelu_ffn, layer_norm, multi_head_attention


def unpack_inputs(token_ids, num_special_tokens=4, bits_per_token=8):
    """:141-163 - one vocabulary slot pair per (function token, bit); rows holding a special token become zeros."""
    lsh = token_ids - num_special_tokens
    bits = torch.cat([((lsh & (1 << hob)) > 0).long() for hob in range(bits_per_token)], dim=1)
    inputs = 2 * torch.arange(bits.shape[1])[None, :] + bits
    special = (token_ids < num_special_tokens).any(dim=1, keepdim=True)
    return torch.where(special, torch.zeros_like(inputs), inputs)


def regression_head(x, w, prefix):
    """esm/layers/regression_head.py:17-22 (hidden_dim from the weights)."""
    y = F.gelu(x @ w[prefix + "0.weight"].T + w[prefix + "0.bias"])
    y = layer_norm(y, w[prefix + "2.weight"], w[prefix + "2.bias"])
    return y @ w[prefix + "3.weight"].T + w[prefix + "3.bias"]


def forward(token_ids, w, n_heads, n_layers):
    x = w["embedding.weight"][unpack_inputs(token_ids)]
    for i in range(n_layers):  # UnifiedTransformerBlock(bias=True, qk_layernorm=False, ffn_type="gelu"), scale 1
        p = f"decoder.blocks.{i}."
        x = x + multi_head_attention(x, None, w, p + "attn.", n_heads)
        x = x + gelu_ffn(x, w, p + "ffn.")
    pooled = layer_norm(x, w["decoder.norm.weight"]).mean(dim=1)
    return {name: regression_head(pooled, w, f"heads.{name}.") for name in ("keyword_logits", "keyword_tfidf", "interpro_logits")}
