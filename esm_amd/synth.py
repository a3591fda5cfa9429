"""Deterministic synthetic weights and inputs (no released checkpoints exist offline).

Weights are drawn per tensor from a Philox counter-based stream keyed by (seed, crc32(name)),
so any subset of tensors can be regenerated identically on another machine without the
reference: matrices ~ N(0, 0.02), LayerNorm gamma ~ N(1, 0.1), biases ~ N(0, 0.1),
geometric-attention per-head scales ~ N(0, 0.5).  Key names and shapes follow the
reference's state_dict (esm/models/esmc.py:54-73, esm/models/esm3.py:61-218).
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def swiglu_hidden(d_model: int, expansion_ratio: float = 8 / 3) -> int:
    # reference: esm/layers/blocks.py:10-12
    return int(((expansion_ratio * d_model) + 255) // 256 * 256)


def _draw(name: str, shape, seed: int, mean: float, std: float) -> torch.Tensor:
    key = (int(seed) << 32) | zlib.crc32(name.encode())
    g = np.random.Generator(np.random.Philox(key=key))
    a = g.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean)
    return torch.from_numpy(a)


def _kind(name: str):
    """(mean, std) for a state_dict key."""
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("_scale_per_head"):
        return 0.0, 0.5
    is_ln = any(s in name for s in ("layernorm_qkv.0.", "q_ln.", "k_ln.", "ffn.0.", "s_norm.", "transformer.norm.")) \
        or (("_head.2." in name) or ("sequence_head.2." in name)) \
        or any(s in name for s in ("affine_output_projection.norm.", "pairwise_classification_head.norm.", "plddt_head.norm.",
                                   "decoder_stack.norm."))
    if is_ln and leaf == "weight":
        return 1.0, 0.1
    if leaf == "bias":
        return 0.0, 0.1
    return 0.0, 0.02


def trunk_shapes(prefix: str, d: int, n_layers: int, v_heads: int | None = None, n_layers_geom: int = 0):
    h = swiglu_hidden(d)
    shapes = {}
    for i in range(n_layers):
        p = f"{prefix}blocks.{i}."
        shapes[p + "attn.layernorm_qkv.0.weight"] = (d,)
        shapes[p + "attn.layernorm_qkv.0.bias"] = (d,)
        shapes[p + "attn.layernorm_qkv.1.weight"] = (3 * d, d)
        shapes[p + "attn.out_proj.weight"] = (d, d)
        shapes[p + "attn.q_ln.weight"] = (d,)
        shapes[p + "attn.k_ln.weight"] = (d,)
        if i < n_layers_geom:
            assert v_heads is not None
            shapes[p + "geom_attn.s_norm.weight"] = (d,)
            shapes[p + "geom_attn.proj.weight"] = (15 * v_heads, d)
            shapes[p + "geom_attn.out_proj.weight"] = (d, 3 * v_heads)
            shapes[p + "geom_attn.distance_scale_per_head"] = (v_heads,)
            shapes[p + "geom_attn.rotation_scale_per_head"] = (v_heads,)
        shapes[p + "ffn.0.weight"] = (d,)
        shapes[p + "ffn.0.bias"] = (d,)
        shapes[p + "ffn.1.weight"] = (2 * h, d)
        shapes[p + "ffn.3.weight"] = (d, h)
    shapes[prefix + "norm.weight"] = (d,)
    return shapes


def head_shapes(prefix: str, d: int, out: int):
    return {
        prefix + "0.weight": (d, d), prefix + "0.bias": (d,),
        prefix + "2.weight": (d,), prefix + "2.bias": (d,),
        prefix + "3.weight": (out, d), prefix + "3.bias": (out,),
    }


def esmc_shapes(d: int, n_layers: int):
    s = {"embed.weight": (64, d)}
    s.update(trunk_shapes("transformer.", d, n_layers))
    s.update(head_shapes("sequence_head.", d, 64))
    return s


ESM3_HEAD_DIMS = {"sequence": 64, "structure": 4096, "ss8": 11, "sasa": 19, "function": 2080, "residue": 1478}


def esm3_shapes(d: int, n_layers: int, v_heads: int):
    s = {
        "encoder.sequence_embed.weight": (64, d),
        "encoder.plddt_projection.weight": (d, 16), "encoder.plddt_projection.bias": (d,),
        "encoder.structure_per_res_plddt_projection.weight": (d, 16),
        "encoder.structure_per_res_plddt_projection.bias": (d,),
        "encoder.structure_tokens_embed.weight": (4101, d),
        "encoder.ss8_embed.weight": (11, d),
        "encoder.sasa_embed.weight": (19, d),
        "encoder.residue_embed.weight": (1478, d),
    }
    for i in range(8):
        s[f"encoder.function_embed.{i}.weight"] = (260, d // 8)
    s.update(trunk_shapes("transformer.", d, n_layers, v_heads=v_heads, n_layers_geom=1))
    for name, out in ESM3_HEAD_DIMS.items():
        s.update(head_shapes(f"output_heads.{name}_head.", d, out))
    return s


def vqdec_shapes(d: int, n_layers: int):
    """StructureTokenDecoder state_dict (esm/models/vqvae.py:325-363)."""
    s = {"embed.weight": (4101, d)}
    s.update(trunk_shapes("decoder_stack.", d, n_layers))
    a, p, q = "affine_output_projection.", "pairwise_classification_head.", "plddt_head."
    s.update({a + "ffn1.weight": (d, d), a + "ffn1.bias": (d,), a + "norm.weight": (d,), a + "norm.bias": (d,),
              a + "proj.weight": (23, d), a + "proj.bias": (23,),
              p + "downproject.weight": (128, d), p + "linear1.weight": (128, 128), p + "norm.weight": (128,),
              p + "norm.bias": (128,), p + "linear2.weight": (224, 128),
              q + "dense.weight": (d, d), q + "dense.bias": (d,), q + "norm.weight": (d,), q + "norm.bias": (d,),
              q + "output.weight": (50, d), q + "output.bias": (50,)})
    return s


def vqenc_shapes(d: int, v_heads: int, n_layers: int, d_out: int = 128, n_codes: int = 4096):
    """StructureTokenEncoder state_dict (esm/models/vqvae.py:179-191; blocks with bias=True, expansion 4)."""
    h = swiglu_hidden(d, 4)
    s = {}
    for i in range(n_layers):
        p = f"transformer.blocks.{i}."
        s.update({p + "geom_attn.s_norm.weight": (d,), p + "geom_attn.s_norm.bias": (d,),
                  p + "geom_attn.proj.weight": (15 * v_heads, d), p + "geom_attn.proj.bias": (15 * v_heads,),
                  p + "geom_attn.out_proj.weight": (d, 3 * v_heads), p + "geom_attn.out_proj.bias": (d,),
                  p + "geom_attn.distance_scale_per_head": (v_heads,), p + "geom_attn.rotation_scale_per_head": (v_heads,),
                  p + "ffn.0.weight": (d,), p + "ffn.0.bias": (d,), p + "ffn.1.weight": (2 * h, d), p + "ffn.1.bias": (2 * h,),
                  p + "ffn.3.weight": (d, h), p + "ffn.3.bias": (d,)})
    s.update({"pre_vq_proj.weight": (d_out, d), "pre_vq_proj.bias": (d_out,), "codebook.embeddings": (n_codes, d_out),
              "codebook.N": (n_codes,), "codebook.z_avg": (n_codes, d_out),
              "relative_positional_embedding.embedding.weight": (66, d)})
    return s


def vqenc_weights(d: int, v_heads: int, n_layers: int, d_out: int = 128, n_codes: int = 4096, seed: int = 0) -> dict:
    """Unit-scale weights for the structure encoder: matrices ~ N(0, 1/fan_in), embeddings and codes ~ N(0, 1).  With
    the N(0, 0.02) matrices used elsewhere the biases swamp the geometry and every residue maps to the same code."""
    out = {}
    for name, shape in vqenc_shapes(d, v_heads, n_layers, d_out, n_codes).items():
        if name.endswith("_scale_per_head"):
            mean, std = 0.0, 0.5
        elif ".s_norm." in name or ".ffn.0." in name:
            mean, std = (1.0, 0.1) if name.endswith("weight") else (0.0, 0.1)
        elif name.endswith(".bias"):
            mean, std = 0.0, 0.1
        elif name.startswith("codebook.") or name.startswith("relative_positional_embedding."):
            mean, std = 0.0, 1.0
        else:
            mean, std = 0.0, float(shape[-1]) ** -0.5
        out[name] = _draw(name, shape, seed, mean, std)
    return out


def fdec_weights(d: int, n_layers: int, n_keywords: int, n_interpro: int, seed: int = 0) -> dict:
    """FunctionTokenDecoder state_dict (esm/models/function_decoder.py:85-125), unit-scale like vqenc_weights."""
    h = 4 * d
    shapes = {"embedding.weight": (128, d), "decoder.norm.weight": (d,)}
    for i in range(n_layers):
        p = f"decoder.blocks.{i}."
        shapes.update({p + "attn.layernorm_qkv.0.weight": (d,), p + "attn.layernorm_qkv.0.bias": (d,),
                       p + "attn.layernorm_qkv.1.weight": (3 * d, d), p + "attn.layernorm_qkv.1.bias": (3 * d,),
                       p + "attn.out_proj.weight": (d, d), p + "attn.out_proj.bias": (d,),
                       p + "ffn.0.weight": (d,), p + "ffn.0.bias": (d,), p + "ffn.1.weight": (h, d), p + "ffn.1.bias": (h,),
                       p + "ffn.3.weight": (d, h), p + "ffn.3.bias": (d,)})
    for name, out in (("keyword_logits", n_keywords), ("keyword_tfidf", n_keywords), ("interpro_logits", n_interpro)):
        p = f"heads.{name}."
        shapes.update({p + "0.weight": (h, d), p + "0.bias": (h,), p + "2.weight": (h,), p + "2.bias": (h,),
                       p + "3.weight": (out, h), p + "3.bias": (out,)})
    out = {}
    for name, shape in shapes.items():
        is_ln = name.endswith("norm.weight") or ".layernorm_qkv.0." in name or ".ffn.0." in name or \
            (name.startswith("heads.") and ".2." in name)
        if is_ln:
            mean, std = (1.0, 0.1) if name.endswith("weight") else (0.0, 0.1)
        elif name.endswith(".bias"):
            mean, std = 0.0, 0.1
        elif name == "embedding.weight":
            mean, std = 0.0, 1.0
        else:
            mean, std = 0.0, float(shape[-1]) ** -0.5
        out[name] = _draw(name, shape, seed, mean, std)
    return out


def stack_variant_weights(shapes: dict, seed: int = 0) -> dict:
    """Unit-scale weights for a TransformerStack of ANY block variant (blocks.py:73-115) from a {key: shape} map -
    normally the module's own state_dict shapes, so the same call works for the reference's module and for ours.
    LayerNorm weights ~ N(1, 0.1), every bias ~ N(0, 0.1), per-head geometric scales ~ N(0, 0.5), matrices
    ~ N(0, 1/fan_in)."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        is_ln = any(t in name for t in (".layernorm_qkv.0.", ".ffn.0.", ".q_ln.", ".k_ln.", ".s_norm.")) or name.endswith("norm.weight")
        if is_ln:
            mean, std = (1.0, 0.1) if name.endswith("weight") else (0.0, 0.1)
        elif name.endswith(".bias"):
            mean, std = 0.0, 0.1
        elif name.endswith("_scale_per_head"):
            mean, std = 0.0, 0.5
        else:
            mean, std = 0.0, float(shape[-1]) ** -0.5
        out[name] = _draw(name, shape, seed, mean, std)
    return out


def make_weights(shapes: dict, seed: int = 0) -> dict:
    out = {}
    for name, shape in shapes.items():
        mean, std = _kind(name)
        t = _draw(name, shape, seed, mean, std)
        # nn.Embedding(padding_idx=0) rows are zero in the reference; keep that invariant so the
        # state_dict round-trips exactly (esm/models/esm3.py:86-90).
        if ("function_embed." in name or "residue_embed." in name) and name.endswith("weight"):
            t[0].zero_()
        out[name] = t
    return out


def esmc_weights(d: int, n_layers: int, seed: int = 0) -> dict:
    return make_weights(esmc_shapes(d, n_layers), seed)


def esm3_weights(d: int, n_layers: int, v_heads: int, seed: int = 0) -> dict:
    return make_weights(esm3_shapes(d, n_layers, v_heads), seed)


# --------------------------------------------------------------------------------------
# synthetic inputs (SURVEY.md section 8d / BASELINE.md section 3)
# --------------------------------------------------------------------------------------
def synth_activations(shape, seed: int, std: float = 1.0) -> torch.Tensor:
    """N(0, std) activations from the same Philox stream as the weights: large test inputs are regenerated from
    their seed on the GPU box instead of being stored in a fixture."""
    return _draw("activations", shape, seed, 0.0, std)


def synth_tokens(lengths, pad_to: int | None = None, seed: int = 1234) -> torch.Tensor:
    """[B, Lmax] int64: <cls> + residues(4..23) + <eos>, right-padded with <pad>=1."""
    lengths = [int(x) for x in lengths]
    lmax = pad_to or max(lengths)
    out = torch.full((len(lengths), lmax), 1, dtype=torch.int64)
    for i, n in enumerate(lengths):
        g = torch.Generator().manual_seed(seed + i)
        row = torch.randint(4, 24, (n,), generator=g)
        row[0] = 0
        row[-1] = 2
        out[i, :n] = row
    return out


def synth_backbone(B: int, L: int, seed: int = 7, nan_frac: float = 0.1, jitter: float = 0.0) -> torch.Tensor:
    """[B, L, 3, 3] fp32 random-walk backbone (CA step 3.8 A), BOS/EOS rows = inf, nan_frac rows NaN.
    jitter > 0 varies the step length by that fraction (no two CA-CA distances tie: nearest-neighbour order unique)."""
    g = torch.Generator().manual_seed(seed)
    step = torch.randn(B, L, 3, generator=g)
    step = 3.8 * step / step.norm(dim=-1, keepdim=True)
    if jitter > 0:
        step = step * (1.0 + jitter * (2 * torch.rand(B, L, 1, generator=torch.Generator().manual_seed(seed + 99991)) - 1))
    ca = torch.cumsum(step, dim=1)
    n = ca + 1.46 * torch.nn.functional.normalize(torch.randn(B, L, 3, generator=g), dim=-1)
    c = ca + 1.52 * torch.nn.functional.normalize(torch.randn(B, L, 3, generator=g), dim=-1)
    coords = torch.stack([n, ca, c], dim=2)
    coords[:, 0] = float("inf")
    coords[:, -1] = float("inf")
    drop = torch.rand(B, L, generator=g) < nan_frac
    coords[drop] = float("nan")
    return coords
