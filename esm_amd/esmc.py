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
from .api import ESMProtein, ESMProteinTensor, ForwardTrackData, LogitsConfig, LogitsOutput, _BatchedESMProteinTensor
from .engine import DeviceBatch, HeadEngine, _stream, segment_mean
from .tokenization import EsmSequenceTokenizer, decode_sequence, stack_variable_length_tensors, tokenize_sequence
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
        self.tokenizer = tokenizer if tokenizer is not None else EsmSequenceTokenizer()
        self.d_model = d_model
        self._head = None
        self._embed_f32 = None
        self._token_err = None  # device counter of invalid tokens / rows seen by the device-described path
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
                       want_logits: bool = True, hidden_layer: int = -1):
        """tokens int64 [N] (packed) -> (logits [N,64]|None, post_norm [N,d], hiddens|None).
        hidden_layer >= 0: capture only that block's output (hiddens [1, N, d])."""
        trunk, head = self._engines()
        N = tokens.shape[0]
        x = torch.empty((N, self.d_model), dtype=torch.float32, device=tokens.device)
        L.check(L.lib().esmk_embed(L.ptr(tokens), L.ptr(self._embed_f32), 64, L.ptr(x), N, self.d_model, _stream()),
                "esmk_embed")
        post, _pre, hid = trunk.forward(x, batch, want_post=True, want_hiddens=want_hiddens, hidden_layer=hidden_layer)
        logits = head.forward(post) if want_logits else None
        return logits, post, hid

    def invalid_token_count(self) -> int:
        """Tokens outside the 64-row table, summed over the forwards run on the device-described path since the model
        was built (synchronises; 0 = every input was valid).  Such tokens are clamped into the table - check this
        where the reference would have raised (or pass validate=True)."""
        return 0 if self._token_err is None else int(self._token_err.item())

    # ---- text <-> tokens (reference esmc.py:100-115,175-192) ----------------------------------
    def _tokenize(self, sequence: list[str]) -> torch.Tensor:
        pad = self.tokenizer.pad_token_id
        return stack_variable_length_tensors(
            [tokenize_sequence(x, self.tokenizer, add_special_tokens=True) for x in sequence],
            constant_value=pad).to(self.device)

    def _detokenize(self, sequence: torch.Tensor) -> list[str]:
        pad = self.tokenizer.pad_token_id
        assert sequence.ndim == 2
        return [decode_sequence(x[x != pad][1:-1], self.tokenizer) for x in sequence]

    def encode(self, input: ESMProtein) -> ESMProteinTensor:
        tokens = self._tokenize([input.sequence])[0] if input.sequence is not None else None
        return ESMProteinTensor(sequence=tokens,
                                potential_sequence_of_concern=input.potential_sequence_of_concern).to(self.device)

    def decode(self, input: ESMProteinTensor) -> ESMProtein:
        assert input.sequence is not None
        return ESMProtein(sequence=self._detokenize(input.sequence[None] if input.sequence.ndim == 1 else input.sequence)[0])

    @torch.no_grad()
    def forward(self, sequence_tokens: torch.Tensor | None = None, sequence_id: torch.Tensor | None = None,
                return_hidden_states: bool = True, _means: dict | None = None, hidden_layer: int = -1,
                drop_pads: bool = False, validate: bool = False) -> ESMCOutput:
        """reference esm/models/esmc.py:117-173.  Beyond the reference's arguments:
        hidden_layer >= 0  capture only that block's output: hidden_states is [1, B, L, d] (LogitsConfig.ith_hidden_layer)
        drop_pads          pack the non-pad tokens on the host first (one host synchronisation; no arithmetic is spent
                           on padding - the form for very ragged batches; esm_amd.pipeline does this for bulk jobs)
        validate           raise IndexError for out-of-range tokens like the reference does (one host synchronisation)

        Default path (sequence_id=None): the batch is described ON THE DEVICE (esmk_compact_tokens) - no host
        synchronisation anywhere in the call.  Any padding pattern is handled as the reference's
        `sequence_id = tokens != pad` defines it (the real tokens of a row attend to each other wherever they sit,
        rotary position = column): real tokens are moved to the front of their row's slots on the way in and back to
        their columns on the way out (nothing moves in right-padded rows); padding positions run through the row-wise
        kernels, are never attended to, and come back as zeros (the reference's flash path, esmc.py:159-164)."""
        assert sequence_tokens is not None and sequence_tokens.ndim == 2
        B, Ln = sequence_tokens.shape
        dev = sequence_tokens.device
        if dev.type != "cuda":
            raise L.EsmkError("esm_amd runs on the GPU only: pass CUDA tensors (there is no CPU path)")
        if validate and (int(sequence_tokens.min()) < 0 or int(sequence_tokens.max()) >= 64):
            raise IndexError("sequence_tokens out of range for the 64-row embedding table")
        flat = sequence_tokens.reshape(-1).to(torch.int64).contiguous()
        NL = self.transformer.n_layers
        n_hid = NL if hidden_layer < 0 else 1
        if sequence_id is None and not drop_pads:
            # ---- device-described batch: no host synchronisation
            if self._token_err is None or self._token_err.device != dev:
                self._token_err = torch.zeros(1, dtype=torch.int32, device=dev)
            # cu [2B+1] | perm [B*L] | positions [B*L] in one int32 buffer; real tokens first in every row's slots
            n = B * Ln
            meta = torch.empty(2 * B + 1 + 2 * n, dtype=torch.int32, device=dev)
            cu, perm, pos = meta[:2 * B + 1], meta[2 * B + 1:2 * B + 1 + n], meta[2 * B + 1 + n:]
            tok = torch.empty_like(flat)
            L.check(L.lib().esmk_compact_tokens(L.ptr(flat), B, Ln, PAD_TOKEN_ID, 64, L.ptr(cu), L.ptr(perm), L.ptr(pos),
                                                L.ptr(tok), L.ptr(self._token_err), _stream()), "esmk_compact_tokens")
            batch = DeviceBatch.padded(B, Ln, cu, dev, positions=pos)
            logits, post, hid = self.forward_packed(tok, batch, want_hiddens=return_hidden_states,
                                                    hidden_layer=hidden_layer)
            if _means is not None:  # means over the valid tokens of every row = the even segments
                if _means.get("embedding"):
                    _means["embedding"] = segment_mean(post, cu)[::2]
                if _means.get("hidden") and hid is not None:
                    _means["hidden"] = torch.stack([segment_mean(hid[i], cu)[::2] for i in range(hid.shape[0])])
            # slots -> columns, zeros at padding (esmc.py:159-164), in place: nothing moves in right-padded rows
            unpack = lambda t, n_mat, w: L.check(L.lib().esmk_unpack_rows(L.ptr(t), n_mat, B, Ln, w, L.ptr(perm), L.ptr(cu),
                                                                         _stream()), "esmk_unpack_rows")
            unpack(logits, 1, logits.shape[-1])
            unpack(post, 1, self.d_model)
            hidden = None
            if hid is not None:
                unpack(hid, n_hid, self.d_model)
                hidden = hid.view(n_hid, B, Ln, self.d_model)
            return ESMCOutput(sequence_logits=logits.view(B, Ln, -1), embeddings=post.view(B, Ln, -1),
                              hidden_states=hidden)
        # ---- host-described batch (explicit sequence_id of any dialect, or drop_pads)
        if sequence_id is None:
            sequence_id = sequence_tokens != PAD_TOKEN_ID  # esmc.py:133-135
        assert sequence_id.shape == (B, Ln)
        pb = pack_sequence_id(sequence_id.detach().cpu().numpy(), B, Ln, drop_false=True)
        batch = DeviceBatch(pb, dev)
        if pb.gather is not None:
            gidx = torch.from_numpy(pb.gather).to(dev)
            tok = flat[gidx].contiguous()
        else:
            gidx, tok = None, flat
        logits, post, hid = self.forward_packed(tok, batch, want_hiddens=return_hidden_states, hidden_layer=hidden_layer)
        if _means is not None:  # per-row means over the kept tokens, on the packed tensors (esmk_segment_mean)
            if _means.get("embedding"):
                _means["embedding"] = segment_mean(post, batch.row_cu)
            if _means.get("hidden") and hid is not None:
                _means["hidden"] = torch.stack([segment_mean(hid[i], batch.row_cu) for i in range(hid.shape[0])])

        def unpack(t):  # [N, w] -> [B, L, w] with zeros at dropped positions
            if gidx is None:
                return t.view(B, Ln, -1)
            out = torch.zeros((B * Ln, t.shape[-1]), dtype=t.dtype, device=dev)
            out[gidx] = t
            return out.view(B, Ln, -1)

        hidden = None
        if hid is not None:
            if gidx is None:
                hidden = hid.view(n_hid, B, Ln, self.d_model)
            else:
                hidden = torch.zeros((n_hid, B * Ln, self.d_model), dtype=hid.dtype, device=dev)
                hidden[:, gidx] = hid
                hidden = hidden.view(n_hid, B, Ln, self.d_model)
        return ESMCOutput(sequence_logits=unpack(logits), embeddings=unpack(post), hidden_states=hidden)

    def logits(self, input: ESMProteinTensor | _BatchedESMProteinTensor,
               config: LogitsConfig = LogitsConfig()) -> LogitsOutput:
        """reference esmc.py:194-225."""
        if not isinstance(input, _BatchedESMProteinTensor):
            input = _BatchedESMProteinTensor.from_protein_tensor(input)
        want_hid = config.return_hidden_states or config.return_mean_hidden_states
        means = {"embedding": config.return_mean_embedding, "hidden": config.return_mean_hidden_states}
        # ith_hidden_layer != -1: only that block's output is ever captured ([1, B, L, d], what the reference's
        # hidden_states[i:i+1] slice returns, esmc.py:213-218); python-style negative indices count from the end
        layer = config.ith_hidden_layer
        if layer != -1 and want_hid:
            if not -self.transformer.n_layers <= layer < self.transformer.n_layers:
                raise IndexError(f"ith_hidden_layer {layer} outside the {self.transformer.n_layers} layers")
            layer %= self.transformer.n_layers
        out = self.forward(sequence_tokens=input.sequence, return_hidden_states=want_hid, _means=means,
                           hidden_layer=layer if want_hid else -1)
        hs, mh = out.hidden_states, means["hidden"] if config.return_mean_hidden_states else None
        return LogitsOutput(
            logits=ForwardTrackData(sequence=out.sequence_logits if config.sequence else None),
            embeddings=out.embeddings if config.return_embeddings else None,
            mean_embedding=means["embedding"] if config.return_mean_embedding else None,
            hidden_states=hs if config.return_hidden_states else None,
            mean_hidden_state=mh)
