"""Import the real reference trunk classes from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY, and it never travels: /root/reference does not exist on the GPU
box, so nothing under tests/ marked gpu, bench.py or smoke() imports this module.  It is used by
`oracle/make_golden.py` to produce the committed fixtures under tests/golden/ and by
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent).

The reference imports 13 third-party packages that are absent here and unused by trunk
arithmetic (zstd, cloudpathlib, biotite, ...).  A meta-path finder serves
inert placeholder modules for exactly those names.  Nothing here may reach
`esm.utils.constants.esm3.data_root` (HF-hub download): no from_pretrained, no
esm.pretrained builders, no get_esm3_model_tokenizers().
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = "/root/reference"
_ABSENT = {"zstd", "cloudpathlib", "biotite", "Bio", "brotli", "msgpack_numpy", "tenacity", "pygtrie",
           "pydssp", "boto3", "ipywidgets", "py3Dmol", "dna_features_viewer"}


class _PlaceholderLoader(importlib.abc.Loader):
    def create_module(self, spec):
        mod = types.ModuleType(spec.name)
        mod.__path__ = []
        mod.__getattr__ = lambda attr, _n=spec.name: MagicMock(name=f"{_n}.{attr}")
        return mod

    def exec_module(self, module):
        return None


class _PlaceholderFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(fullname, _PlaceholderLoader(), is_package=True)
        return None


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "esm"))


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.meta_path.insert(0, _PlaceholderFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def build_esmc(d_model: int, n_heads: int, n_layers: int, weights: dict):
    install()
    from esm.models.esmc import ESMC
    from esm.tokenization.sequence_tokenizer import EsmSequenceTokenizer

    m = ESMC(d_model=d_model, n_heads=n_heads, n_layers=n_layers, tokenizer=EsmSequenceTokenizer(),
             use_flash_attn=False).eval()
    missing, unexpected = m.load_state_dict(weights, strict=True)
    assert not missing and not unexpected
    return m


def build_esm3(d_model: int, n_heads: int, v_heads: int, n_layers: int, weights: dict):
    install()
    from esm.models.esm3 import ESM3
    from esm.tokenization.sequence_tokenizer import EsmSequenceTokenizer

    m = ESM3(d_model=d_model, n_heads=n_heads, v_heads=v_heads, n_layers=n_layers,
             structure_encoder_fn=None, structure_decoder_fn=None, function_decoder_fn=None,
             tokenizers=types.SimpleNamespace(sequence=EsmSequenceTokenizer())).eval()
    m.load_state_dict(weights, strict=True)
    return m


def build_stack(d_model, n_heads, n_layers, weights: dict, v_heads=None, n_layers_geom=0,
                mask_and_zero_frameless=False, scale_residue=True):
    install()
    from esm.layers.transformer_stack import TransformerStack

    m = TransformerStack(d_model, n_heads, v_heads, n_layers, n_layers_geom=n_layers_geom,
                         mask_and_zero_frameless=mask_and_zero_frameless, scale_residue=scale_residue).eval()
    m.load_state_dict(weights, strict=True)
    return m


def build_stack_variant(d_model, n_heads, n_layers, v_heads=None, **kw):
    """The reference's TransformerStack with any constructor options (weights are loaded by the caller)."""
    install()
    from esm.layers.transformer_stack import TransformerStack

    return TransformerStack(d_model, n_heads, v_heads, n_layers, **kw).eval()


def build_vqdec(d_model, n_heads, n_layers, weights: dict):
    install()
    from esm.models.vqvae import StructureTokenDecoder

    m = StructureTokenDecoder(d_model=d_model, n_heads=n_heads, n_layers=n_layers).eval()
    m.load_state_dict(weights, strict=True)
    return m


def build_vqenc(d_model, v_heads, n_layers, d_out, n_codes, weights: dict):
    install()
    from esm.models.vqvae import StructureTokenEncoder

    m = StructureTokenEncoder(d_model=d_model, n_heads=1, v_heads=v_heads, n_layers=n_layers, d_out=d_out,
                              n_codes=n_codes).eval()
    m.load_state_dict(weights, strict=True)
    return m


def build_fdec(d_model, n_heads, n_layers, n_keywords, n_interpro, weights: dict):
    """The reference FunctionTokenDecoder WITHOUT its constructor: that reads the InterPro entry list and the keyword
    vocabulary through data_root (a download).  The sub-modules below are exactly the ones the constructor builds
    (function_decoder.py:85-125); forward() is the reference's own."""
    install()
    import torch.nn as nn
    from esm.layers.regression_head import RegressionHead
    from esm.layers.transformer_stack import TransformerStack
    from esm.models.function_decoder import FunctionTokenDecoder, FunctionTokenDecoderConfig

    cfg = FunctionTokenDecoderConfig(d_model=d_model, n_heads=n_heads, n_layers=n_layers, num_interpro_classes=n_interpro,
                                     keyword_vocabulary_size=n_keywords, interpro_entry_list="", keyword_vocabulary_path="")
    m = FunctionTokenDecoder.__new__(FunctionTokenDecoder)
    nn.Module.__init__(m)
    m.config = cfg
    m.embedding = nn.Embedding(2 * cfg.function_token_depth * cfg.bits_per_token, d_model)
    m.decoder = TransformerStack(d_model=d_model, n_heads=n_heads, v_heads=None, n_layers=n_layers, n_layers_geom=0,
                                 scale_residue=False, bias=True, qk_layernorm=False, ffn_type="gelu", expansion_ratio=4)
    m.heads = nn.ModuleDict({k: RegressionHead(d_model=d_model, output_dim=o, hidden_dim=4 * d_model)
                             for k, o in (("keyword_logits", n_keywords), ("keyword_tfidf", n_keywords),
                                          ("interpro_logits", n_interpro))})
    m.load_state_dict(weights, strict=True)
    return m.eval()


def build_frames(coords):
    install()
    from esm.utils.structure.affine3d import build_affine3d_from_coordinates

    affine, mask = build_affine3d_from_coordinates(coords)
    return affine, mask
