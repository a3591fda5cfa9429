"""ESM C with the reference's Python surface (reference esm/models/esmc.py:37-225) on the HIP trunk.

forward(sequence_tokens, sequence_id=None) -> ESMCOutput(sequence_logits, embeddings, hidden_states):
embed -> drop pads (the reference's own flash-path formulation, esmc.py:142-164) -> stack -> head.
Pad positions come back as zeros, as in the reference's flash path.  Outputs are fp32.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .api import ESMProteinTensor, ForwardTrackData, LogitsConfig, LogitsOutput, _BatchedESMProteinTensor
from .engine import DeviceBatch, HeadEngine, _stream
from .packing import PackedBatch, pack_sequence_id
from .transformer_stack import TransformerStack

PAD_TOKEN_ID = 1  # esm/utils/constants/esm3.py:8


@dataclass
class ESMCOutput:
    sequence_logits: torch.Tensor
    embeddings: torch.Tensor | None
    hidden_states: torch.Tensor | None


def RegressionHeadParams(d_model: int, output_dim: int) -> nn.Module:
    """Parameter container with the reference's RegressionHead keys (0, 2, 3)."""
    return nn.Sequential(nn.Linear(d_model, d_model), nn.Identity(), nn.LayerNorm(d_model),
                         nn.Linear(d_model, output_dim))


class ESMC(nn.Module):
    def __init__(self, d_model: int, n_heads: int, n_layers: int, tokenizer=None, use_flash_attn: bool = True):
        super().__init__()
        self.embed = nn.Embedding(64, d_model)
        self.transformer = TransformerStack(d_model, n_heads, None, n_layers, n_layers_geom=0)
        self.sequence_head = RegressionHeadParams(d_model, 64)
        self.tokenizer = tokenizer
        self.d_model = d_model
        self._head = None
        self._embed_f32 = None
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._head = None
        self._embed_f32 = None

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.embed.weight.device

    @property
    def raw_model(self):
        return self

    def _engines(self):
        if self._head is None:
            dev = self.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the model to cuda (there is no CPU path)")
            sd = {k: v for k, v in self.state_dict().items() if k.startswith("sequence_head.")}
            self._head = HeadEngine(sd, "sequence_head.", self.d_model, 64, dev)
            self._embed_f32 = self.embed.weight.detach().to(torch.float32).contiguous()
        return self.transformer.engine(), self._head

    # ------------------------------------------------------------------------------------------
    def forward_packed(self, tokens: torch.Tensor, batch: DeviceBatch, want_hiddens: bool = False,
                       want_logits: bool = True):
        """tokens int64 [N] (packed, pads already dropped) -> (logits [N,64]|None, post_norm [N,d], hiddens|None)."""
        trunk, head = self._engines()
        N = tokens.shape[0]
        x = torch.empty((N, self.d_model), dtype=torch.float32, device=tokens.device)
        L.check(L.lib().esmk_embed(L.ptr(tokens), L.ptr(self._embed_f32), 64, L.ptr(x), N, self.d_model, _stream()),
                "esmk_embed")
        post, _pre, hid = trunk.forward(x, batch, want_post=True, want_hiddens=want_hiddens)
        logits = head.forward(post) if want_logits else None
        return logits, post, hid

    @torch.no_grad()
    def forward(self, sequence_tokens: torch.Tensor | None = None, sequence_id: torch.Tensor | None = None,
                return_hidden_states: bool = True) -> ESMCOutput:
        assert sequence_tokens is not None and sequence_tokens.ndim == 2
        B, Ln = sequence_tokens.shape
        dev = sequence_tokens.device
        if int(sequence_tokens.min()) < 0 or int(sequence_tokens.max()) >= 64:
            raise IndexError("sequence_tokens out of range for the 64-row embedding table")
        if sequence_id is None:
            sequence_id = sequence_tokens != PAD_TOKEN_ID  # esmc.py:133-135
        assert sequence_id.shape == (B, Ln)
        pb = pack_sequence_id(sequence_id.detach().cpu().numpy(), B, Ln, drop_false=True)
        batch = DeviceBatch(pb, dev)
        flat = sequence_tokens.reshape(-1).to(torch.int64)
        if pb.gather is not None:
            gidx = torch.from_numpy(pb.gather).to(dev)
            tok = flat[gidx].contiguous()
        else:
            gidx, tok = None, flat.contiguous()
        logits, post, hid = self.forward_packed(tok, batch, want_hiddens=return_hidden_states)

        def unpack(t):  # [N, w] -> [B, L, w] with zeros at dropped positions
            if gidx is None:
                return t.view(B, Ln, -1)
            out = torch.zeros((B * Ln, t.shape[-1]), dtype=t.dtype, device=dev)
            out[gidx] = t
            return out.view(B, Ln, -1)

        hidden = None
        if hid is not None:
            if gidx is None:
                hidden = hid.view(self.transformer.n_layers, B, Ln, self.d_model)
            else:
                hidden = torch.zeros((self.transformer.n_layers, B * Ln, self.d_model), dtype=hid.dtype, device=dev)
                hidden[:, gidx] = hid
                hidden = hidden.view(self.transformer.n_layers, B, Ln, self.d_model)
        return ESMCOutput(sequence_logits=unpack(logits), embeddings=unpack(post), hidden_states=hidden)

    def logits(self, input: ESMProteinTensor | _BatchedESMProteinTensor,
               config: LogitsConfig = LogitsConfig()) -> LogitsOutput:
        """reference esmc.py:194-225."""
        if not isinstance(input, _BatchedESMProteinTensor):
            input = _BatchedESMProteinTensor.from_protein_tensor(input)
        out = self.forward(sequence_tokens=input.sequence, return_hidden_states=config.return_hidden_states)
        hs = out.hidden_states
        if hs is not None and config.ith_hidden_layer != -1:
            hs = hs[config.ith_hidden_layer: config.ith_hidden_layer + 1]
        return LogitsOutput(
            logits=ForwardTrackData(sequence=out.sequence_logits if config.sequence else None),
            embeddings=out.embeddings if config.return_embeddings else None,
            hidden_states=hs if config.return_hidden_states else None)
