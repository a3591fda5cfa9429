"""ESM3 forward / logits with the reference's Python surface (reference esm/models/esm3.py:50-178,
260-374,503-559) on the HIP trunk: 8-track input encoder, backbone frames, 48-block stack whose first
block has geometric attention, six regression heads; `encode` / `decode` for the sequence and structure tracks over
the VQ-VAE models of esm_amd/vqvae.py (esm3.py:414-501, utils/encoding.py:48-97, utils/decoding.py:28-169).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib as L
from .affine3d import build_affine3d_from_coordinates
from .api import ESMProtein, ESMProteinTensor, ForwardTrackData, LogitsConfig, LogitsOutput, _BatchedESMProteinTensor
from .engine import DeviceBatch, HeadEngine, _f32, _stream
from .esmc import RegressionHeadParams
from .transformer_stack import TransformerStack

# esm/utils/constants/esm3.py:7-39
SEQUENCE_BOS_TOKEN, SEQUENCE_PAD_TOKEN, SEQUENCE_EOS_TOKEN, SEQUENCE_CHAINBREAK_TOKEN, SEQUENCE_MASK_TOKEN = 0, 1, 2, 31, 32
STRUCTURE_MASK_TOKEN, STRUCTURE_EOS_TOKEN, STRUCTURE_BOS_TOKEN, STRUCTURE_PAD_TOKEN, STRUCTURE_CHAINBREAK_TOKEN = \
    4096, 4097, 4098, 4099, 4100
SS8_PAD_TOKEN = SASA_PAD_TOKEN = INTERPRO_PAD_TOKEN = RESIDUE_PAD_TOKEN = 0

_HEADS = (("sequence_head", 64), ("structure_head", 4096), ("ss8_head", 11), ("sasa_head", 19),
          ("function_head", 260 * 8), ("residue_head", 1478))


@dataclass
class ESMOutput:
    sequence_logits: torch.Tensor
    structure_logits: torch.Tensor
    secondary_structure_logits: torch.Tensor
    sasa_logits: torch.Tensor
    function_logits: torch.Tensor
    residue_logits: torch.Tensor
    embeddings: torch.Tensor


class EncodeInputs(nn.Module):
    """Parameter container with the reference's keys (esm3.py:69-90); arithmetic in esmk_esm3_encode."""

    def __init__(self, d_model: int):
        super().__init__()
        self.sequence_embed = nn.Embedding(64, d_model)
        self.plddt_projection = nn.Linear(16, d_model)
        self.structure_per_res_plddt_projection = nn.Linear(16, d_model)
        self.structure_tokens_embed = nn.Embedding(4096 + 5, d_model)
        self.ss8_embed = nn.Embedding(8 + 3, d_model)
        self.sasa_embed = nn.Embedding(16 + 3, d_model)
        self.function_embed = nn.ModuleList([nn.Embedding(260, d_model // 8, padding_idx=0) for _ in range(8)])
        self.residue_embed = nn.EmbeddingBag(1478, d_model, mode="sum", padding_idx=0)


class OutputHeads(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        for name, dim in _HEADS:
            setattr(self, name, RegressionHeadParams(d_model, dim))


class ESM3(nn.Module):
    def __init__(self, d_model: int, n_heads: int, v_heads: int, n_layers: int, structure_encoder_fn=None,
                 structure_decoder_fn=None, function_decoder_fn=None, tokenizers=None):
        super().__init__()
        self.encoder = EncodeInputs(d_model)
        self.transformer = TransformerStack(d_model, n_heads, v_heads, n_layers, mask_and_zero_frameless=True)
        self.output_heads = OutputHeads(d_model)
        self.tokenizers = tokenizers or SimpleNamespace(sequence=SimpleNamespace(mask_token_id=SEQUENCE_MASK_TOKEN))
        self.d_model = d_model
        # builders `fn(device) -> model`, as in the reference (esm3.py:181-258); built lazily
        self.structure_encoder_fn, self.structure_decoder_fn = structure_encoder_fn, structure_decoder_fn
        self.function_decoder_fn = function_decoder_fn
        self._structure_encoder = self._structure_decoder = self._function_decoder = None
        self._enc = None
        self._token_err = None  # device counter of out-of-table ids seen by forward()
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._enc = None

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.encoder.sequence_embed.weight.device

    @property
    def raw_model(self):
        return self

    def _engines(self):
        if self._enc is None:
            dev = self.device
            if dev.type != "cuda":
                raise L.EsmkError("esm_amd runs on the GPU only: move the model to cuda (there is no CPU path)")
            e = self.encoder
            keep = [
                _f32(e.sequence_embed.weight, dev), _f32(e.plddt_projection.weight, dev),
                _f32(e.plddt_projection.bias, dev), _f32(e.structure_per_res_plddt_projection.weight, dev),
                _f32(e.structure_per_res_plddt_projection.bias, dev), _f32(e.structure_tokens_embed.weight, dev),
                _f32(e.ss8_embed.weight, dev), _f32(e.sasa_embed.weight, dev),
                torch.stack([_f32(m.weight, dev) for m in e.function_embed]).contiguous(),
                _f32(e.residue_embed.weight, dev),
            ]
            desc = L.EsmkEsm3Encoder(*[L.ptr(t) for t in keep])
            sd = dict(self.output_heads.state_dict())
            heads = {name: HeadEngine(sd, name + ".", self.d_model, dim, dev) for name, dim in _HEADS}
            self._enc = (desc, keep, heads)
        return self._enc

    @torch.no_grad()
    def forward(self, *, sequence_tokens=None, structure_tokens=None, ss8_tokens=None, sasa_tokens=None,
                function_tokens=None, residue_annotation_tokens=None, average_plddt=None, per_res_plddt=None,
                structure_coords=None, chain_id=None, sequence_id=None, _heads=None, validate: bool = False) -> ESMOutput:
        """reference esm3.py:299-374.  _heads (private): names of the output heads to evaluate (default all six);
        the sampling loop asks for one.

        The missing-track defaults, the batch-1 broadcasting and the structure-token fix-ups of esm3.py:317-359 happen
        inside `esmk_esm3_encode_ex`; with sequence_id=None (what `logits()` and the sampler pass) nothing in this call
        waits for the GPU.  Ids outside their embedding tables are clamped and COUNTED on the device
        (`invalid_token_count()`); validate=True checks them on the host first and raises IndexError (one synchronisation)."""
        import ctypes as C
        tracks = (("sequence_tokens", sequence_tokens, 64, ()), ("structure_tokens", structure_tokens, 4101, ()),
                  ("ss8_tokens", ss8_tokens, 11, ()), ("sasa_tokens", sasa_tokens, 19, ()),
                  ("function_tokens", function_tokens, 260, (8,)),
                  ("residue_annotation_tokens", residue_annotation_tokens, 1478, (16,)))
        shaped = [t for _n, t, _v, _tail in tracks if t is not None] + ([structure_coords] if structure_coords is not None else [])
        if not shaped:
            raise ValueError("At least one of the inputs must be non-None")
        Ln, device = shaped[0].shape[1], shaped[0].device
        B = max(t.shape[0] for t in shaped + [t for t in (chain_id, sequence_id, per_res_plddt) if t is not None])
        if device.type != "cuda":
            raise L.EsmkError("esm_amd runs on the GPU only: pass CUDA tensors (there is no CPU path)")
        if validate:
            for name, t, vocab, _tail in tracks:
                if t is None:
                    continue
                lo, hi = int(t.min()), int(t.max())
                if hi >= vocab or lo < (-1 if name == "structure_tokens" else 0):
                    raise IndexError(f"{name} out of range [0, {vocab})")
        desc, _keep, heads = self._engines()
        if self._token_err is None or self._token_err.device != device:
            self._token_err = torch.zeros(1, dtype=torch.int32, device=device)
        M = B * Ln
        inp = L.EsmkEsm3Inputs()
        live = []  # keeps converted tensors alive until the launch is enqueued

        def put(i, field, t, dtype, tail):
            if t is None:
                return
            if t.ndim == 0:  # a scalar (reference ESM3.logits passes average_plddt = tensor(1.0)): one row, broadcast
                t = t.reshape(1, 1).expand(1, Ln)
            if tuple(t.shape[1:]) != (Ln, *tail) or t.shape[0] not in (1, B):
                raise ValueError(f"{field}: expected shape [1 or {B}, {Ln}{''.join(', %d' % x for x in tail)}], got {tuple(t.shape)}")
            t = t.to(device=device, dtype=dtype).contiguous()
            live.append(t)
            setattr(inp, field, L.ptr(t))
            inp.rows[i] = t.shape[0]

        for i, ((_name, t, _v, tail), field) in enumerate(zip(tracks, ("sequence", "structure", "ss8", "sasa",
                                                                        "function_tokens", "residue_tokens"))):
            put(i, field, t, torch.int64, tail)
        put(6, "average_plddt", average_plddt, torch.float32, ())
        put(7, "per_res_plddt", per_res_plddt, torch.float32, ())
        x = torch.empty((M, self.d_model), dtype=torch.float32, device=device)
        L.check(L.lib().esmk_esm3_encode_ex(C.byref(desc), C.byref(inp), L.ptr(x), B, Ln, self.d_model,
                                            L.ptr(self._token_err), _stream()), "esmk_esm3_encode_ex")
        # backbone frames (affine3d.py:512-560); no coordinates = no frames anywhere (esm3.py:338-341 fills NaN)
        if structure_coords is None:
            structure_coords = torch.full((1, Ln, 3, 3), float("nan"), dtype=torch.float32, device=device)
        coords = structure_coords[..., :3, :]
        if coords.shape[0] != B:
            coords = coords.expand(B, Ln, 3, 3)
        affine, affine_mask = build_affine3d_from_coordinates(coords)
        if sequence_id is None:
            # one segment per row, every position attends everywhere (attention.py:73-78): the batch description
            # depends on (B, L) alone and is cached on the device - no host work, no synchronisation
            ch = None if chain_id is None else chain_id.expand(B, Ln).reshape(M).to(torch.int32).contiguous()
            batch = DeviceBatch.rows(B, Ln, device, affine.tensor.reshape(M, 12), affine_mask.reshape(M).to(torch.uint8), ch)
            post, pre, _ = self.transformer.engine().forward(x, batch, want_post=True, want_hiddens=False)
            post, pre = post.view(B, Ln, -1), pre.view(B, Ln, -1)
        else:
            cid = torch.zeros((1, Ln), dtype=torch.int64, device=device) if chain_id is None else chain_id
            post, pre, _ = self.transformer(x.view(B, Ln, self.d_model), sequence_id.expand(B, Ln).contiguous(), affine,
                                            affine_mask, cid.expand(B, Ln).contiguous(), return_hiddens=False)
        pn = post.reshape(M, self.d_model)
        out = {name: (heads[name].forward(pn).view(B, Ln, -1) if _heads is None or name in _heads else None)
               for name, _ in _HEADS}
        fl = out["function_head"]
        return ESMOutput(
            sequence_logits=out["sequence_head"], structure_logits=out["structure_head"],
            secondary_structure_logits=out["ss8_head"], sasa_logits=out["sasa_head"],
            function_logits=None if fl is None else fl.view(B, Ln, 8, 260), residue_logits=out["residue_head"],
            embeddings=pre)

    def invalid_token_count(self) -> int:
        """Ids outside their embedding tables seen by forward() since the model was built (synchronises; 0 = none).
        They were clamped into the table - where the reference's embedding lookup would have failed."""
        return 0 if self._token_err is None else int(self._token_err.item())

    def logits(self, input: ESMProteinTensor | _BatchedESMProteinTensor,
               config: LogitsConfig = LogitsConfig()) -> LogitsOutput:
        """reference esm3.py:503-559."""
        if not isinstance(input, _BatchedESMProteinTensor):
            input = _BatchedESMProteinTensor.from_protein_tensor(input)
        per_res_plddt = None
        if input.coordinates is not None:
            per_res_plddt = input.coordinates.isfinite().all(dim=-1).any(dim=-1).float()
        want = {"sequence_head": config.sequence, "structure_head": config.structure,
                "ss8_head": config.secondary_structure, "sasa_head": config.sasa, "function_head": config.function,
                "residue_head": config.residue_annotations}
        o = self.forward(_heads={k for k, v in want.items() if v},sequence_tokens=input.sequence, structure_tokens=input.structure,
                         ss8_tokens=input.secondary_structure, sasa_tokens=input.sasa, function_tokens=input.function,
                         residue_annotation_tokens=input.residue_annotations,
                         average_plddt=None,  # the reference passes 1.0 here, which is also forward()'s default
                         per_res_plddt=per_res_plddt,
                         structure_coords=input.coordinates, chain_id=None, sequence_id=None)
        return LogitsOutput(
            logits=ForwardTrackData(
                sequence=o.sequence_logits if config.sequence else None,
                structure=o.structure_logits if config.structure else None,
                secondary_structure=o.secondary_structure_logits if config.secondary_structure else None,
                sasa=o.sasa_logits if config.sasa else None,
                function=o.function_logits if config.function else None),
            residue_annotation_logits=o.residue_logits if config.residue_annotations else None,
            embeddings=o.embeddings if config.return_embeddings else None)

    # ---- iterative decoding of one track (reference ESM3.generate -> iterative_sampling_tokens,
    #      esm3.py:417-470, utils/generation.py:355-534) --------------------------------------------
    _TRACK_IDS = {  # track -> (BOS, EOS, PAD, MASK) ids used to fill a missing track (esm3 constants :7-31)
        "sequence": (0, 2, 1, 32), "structure": (4098, 4097, 4099, 4096),
        "secondary_structure": (0, 0, 0, 0), "sasa": (0, 0, 0, 0)}

    def generate_tokens(self, prompts, config, generator=None):
        """Fill the masked positions of `config.track` for a list of ESMProteinTensor prompts (any lengths).

        Other tracks present in a prompt condition the model and are returned unchanged; a missing target track
        starts fully masked.  Returns a list of ESMProteinTensor with the decoded track."""
        from .sampling import default_track_tokenizers, iterative_sampling
        if isinstance(prompts, ESMProteinTensor):
            prompts = [prompts]
        track = config.track
        if track in ("coordinates", "residue_annotations"):
            raise ValueError(f"Iterative sampling {track} is not supported.")
        if track not in self._TRACK_IDS:
            raise NotImplementedError(f"track {track!r} cannot be generated here (needs tokenizer data files)")
        dev = self.device
        lens = [len(p) for p in prompts]
        Ln = max(lens)
        fields = ("sequence", "structure", "secondary_structure", "sasa")

        def stack(name):
            rows = [getattr(p, name) for p in prompts]
            if name == "structure" and config.condition_on_coordinates_only:
                # generation.py:367-370: a prompt with coordinates is conditioned on them alone - its structure tokens
                # are dropped (and decoded from scratch when structure is the target track)
                rows = [None if p.coordinates is not None else r for p, r in zip(prompts, rows)]
            if all(r is None for r in rows) and name != track:
                return None
            bos, eos, pad, mask = self._TRACK_IDS[name]
            out = torch.full((len(prompts), Ln), pad, dtype=torch.int64, device=dev)
            for i, (r, n) in enumerate(zip(rows, lens)):
                if r is None:
                    out[i, :n] = mask
                    out[i, 0], out[i, n - 1] = bos, eos
                else:
                    out[i, :n] = r.to(dev)
            return out

        tracks = {name: stack(name) for name in fields}
        coords = None
        if any(p.coordinates is not None for p in prompts):
            coords = torch.full((len(prompts), Ln, 3, 3), float("nan"), device=dev)
            for i, p in enumerate(prompts):
                if p.coordinates is not None:
                    coords[i, :lens[i]] = p.coordinates[..., :3, :].to(dev)
        cfg_field = {"sequence": "sequence", "structure": "structure", "secondary_structure": "secondary_structure", "sasa": "sasa"}[track]

        def logits_fn(cur):
            t = dict(tracks)
            t[track] = cur
            inp = _BatchedESMProteinTensor(sequence=t["sequence"], structure=t["structure"],
                                           secondary_structure=t["secondary_structure"], sasa=t["sasa"], coordinates=coords)
            return getattr(self.logits(inp, LogitsConfig(**{cfg_field: True})).logits, track)

        tok = default_track_tokenizers()[track]
        done = iterative_sampling(logits_fn, tracks[track], lens, config, tok, generator=generator)
        outs = []
        for i, p in enumerate(prompts):
            q = ESMProteinTensor(**{f.name: getattr(p, f.name) for f in __import__("dataclasses").fields(ESMProteinTensor)})
            setattr(q, track, done[i, :lens[i]].clone())
            outs.append(q)
        return outs

    def generate(self, input, config):
        """esm3.py:377-381: one prompt through batch_generate."""
        proteins = self.batch_generate([input], [config])
        assert len(proteins) == 1
        return proteins[0]

    def batch_generate(self, inputs: list, configs: list, generator=None) -> list:
        """esm3.py:383-412 with iterative_sampling_raw / iterative_sampling_tokens (utils/generation.py:99-127,
        355-534).  ESMProtein prompts are encoded, decoded as tokens and turned back into ESMProtein;
        ESMProteinTensor prompts come back as ESMProteinTensor.  Every prompt has its own GenerationConfig: like the
        reference, a prompt with fewer masked positions than `num_steps` gets its `config.num_steps` lowered to that
        number (in place).  Prompts whose configs then agree share one batched decoding loop (one forward and one
        sampling kernel per step for the whole group; the reference loops over prompts in Python at every step).
        A prompt that cannot be decoded - `coordinates` / `residue_annotations` track, no masked position - yields an
        ESMProteinError(500) in its slot and does not disturb the others (generation.py:423-431,493-495)."""
        import dataclasses

        from .api import ESMProteinError
        assert len(inputs) == len(configs), "Must have the same number of prompts and configs."
        if inputs == []:
            return []
        t = type(inputs[0])
        for x in inputs[1:]:
            assert isinstance(x, t), f"Prompts must have the same type. Got {t.__name__} and {type(x).__name__} instead."
        if isinstance(inputs[0], ESMProtein):
            toks = [self.encode(p) for p in inputs]
            outs = self.batch_generate(toks, configs, generator=generator)
            return [o if isinstance(o, ESMProteinError) else self.decode(o) for o in outs]
        if not isinstance(inputs[0], ESMProteinTensor):
            raise ValueError("Input must be an ESMProtein or ESMProteinTensor")
        devices = {p.device for p in inputs}
        if len(devices) > 1:
            raise AttributeError(f"Input tokens on multiple devices {devices}")
        from .sampling import default_track_tokenizers
        toks = default_track_tokenizers()
        results: list = [None] * len(inputs)
        groups: dict = {}
        for i, (p, cfg) in enumerate(zip(inputs, configs)):
            track = cfg.track
            if track in ("coordinates", "residue_annotations"):
                results[i] = ESMProteinError(500, f"Iterative sampling {track} is not supported.")
                continue
            if track not in self._TRACK_IDS:
                results[i] = ESMProteinError(500, f"track {track!r} cannot be generated here (needs tokenizer data files)")
                continue
            cur = getattr(p, track)
            if track == "structure" and cfg.condition_on_coordinates_only and p.coordinates is not None:
                cur = None
            if cur is None:  # the whole track is decoded: every non-special sequence position (generation.py:333-347)
                if p.sequence is None:
                    n_masks = len(p) - 2
                else:
                    special = [x for x in toks["sequence"].special_token_ids if x != SEQUENCE_MASK_TOKEN]
                    n_masks = int((~torch.isin(p.sequence.cpu(), torch.tensor(special))).sum())
            else:
                m = cur == toks[track].mask_token_id
                m[0], m[-1] = False, False
                n_masks = int(m.sum())
            if n_masks == 0:
                results[i] = ESMProteinError(500, f"Cannot sample {track} when input has no masks.")
                continue
            if n_masks < cfg.num_steps:
                cfg.num_steps = n_masks
            key = tuple((f.name, tuple(v) if isinstance(v, (list, tuple)) else v)
                        for f in dataclasses.fields(cfg) for v in [getattr(cfg, f.name)])
            groups.setdefault(key, []).append(i)
        for idx in groups.values():
            done = self.generate_tokens([inputs[i] for i in idx], configs[idx[0]], generator=generator)
            for i, d in zip(idx, done):
                results[i] = d
        return results

    # ---- structure / function models (esm3.py:245-258) ---------------------------------------------
    def _built(self, slot, fn, what):
        if getattr(self, slot) is None:
            if fn is None:
                raise ValueError(f"this ESM3 was constructed without a {what} builder ({what}_fn)")
            setattr(self, slot, fn(self.device))
        return getattr(self, slot)

    def get_structure_encoder(self):
        return self._built("_structure_encoder", self.structure_encoder_fn, "structure_encoder")

    def get_structure_decoder(self):
        return self._built("_structure_decoder", self.structure_decoder_fn, "structure_decoder")

    def get_function_decoder(self):
        return self._built("_function_decoder", self.function_decoder_fn, "function_decoder")

    # ---- ESMProtein <-> ESMProteinTensor for the sequence and structure tracks ----------------------
    @torch.no_grad()
    def encode(self, input: ESMProtein) -> ESMProteinTensor:
        """esm3.py:414-493 for `sequence`, `secondary_structure`, `sasa` and `coordinates` (function annotations need
        the InterPro tokenizers' data files).  coordinates [L, >=3, 3] with
        N, CA, C first (atom37 works), NaN where unknown -> structure tokens from the VQ-VAE encoder, BOS / EOS added,
        coordinates padded with inf rows (utils/encoding.py:48-97)."""
        from .tokenization import (EsmSequenceTokenizer, SASADiscretizingTokenizer, SecondaryStructureTokenizer,
                                   tokenize_sasa, tokenize_secondary_structure, tokenize_sequence)
        dev = self.device
        seq_tok = struct_tok = coords = ss_tok = sasa_tok = None
        if input.secondary_structure is not None:
            ss_tok = tokenize_secondary_structure(input.secondary_structure, SecondaryStructureTokenizer()).to(dev)
        if input.sasa is not None:
            sasa_tok = tokenize_sasa(input.sasa, SASADiscretizingTokenizer()).to(dev)
        if input.sequence is not None:
            tok = getattr(self.tokenizers, "sequence", None)
            if not hasattr(tok, "encode"):
                tok = EsmSequenceTokenizer()
            seq_tok = tokenize_sequence(input.sequence, tok, add_special_tokens=True).to(dev)
        if input.coordinates is not None:
            c = input.coordinates.to(dev, torch.float32)
            if input.sequence and len(input.sequence) != c.size(0):
                raise ValueError(f"Reference sequence length ({len(input.sequence)}) does not match the number of residues "
                                 f"in the coordinates ({c.size(0)})")
            res = torch.arange(c.size(0), device=dev)[None]
            _, t = self.get_structure_encoder().encode(c[None], residue_index=res)
            struct_tok = torch.nn.functional.pad(t[0], (1, 1), value=STRUCTURE_MASK_TOKEN)
            struct_tok[0], struct_tok[-1] = STRUCTURE_BOS_TOKEN, STRUCTURE_EOS_TOKEN
            coords = torch.nn.functional.pad(c, (0, 0, 0, 0, 1, 1), value=torch.inf)
        if seq_tok is None and struct_tok is None and ss_tok is None and sasa_tok is None:
            raise ValueError("Cannot infer input length from input data. Please provide one of: sequence, structure, "
                             "secondary_structure, sasa.")
        return ESMProteinTensor(sequence=seq_tok, structure=struct_tok, secondary_structure=ss_tok, sasa=sasa_tok,
                                coordinates=coords, potential_sequence_of_concern=input.potential_sequence_of_concern)

    @torch.no_grad()
    def decode(self, input: ESMProteinTensor) -> ESMProtein:
        """utils/decoding.py:28-111 for the sequence, secondary-structure, SASA and structure tracks: the structure tokens (when none of them is
        MASK) go through the VQ-VAE decoder and take priority over `coordinates`.  coordinates come back as the
        backbone N, CA, C [L, 3, 3]: all-atom reconstruction (`ProteinChain.infer_oxygen`) is out of scope."""
        from .tokenization import (EsmSequenceTokenizer, SASADiscretizingTokenizer, SecondaryStructureTokenizer,
                                   decode_sasa, decode_secondary_structure, decode_sequence)
        from .vqvae import decode_structure
        if input.function is not None or input.residue_annotations is not None:
            raise NotImplementedError("function / residue annotations need the InterPro vocabularies (data files)")
        sequence = coords = plddt = ptm = pae = secondary_structure = sasa = None
        # a track whose body is all <pad> decodes to None (decoding.py:46-56)
        if input.secondary_structure is not None and not bool((input.secondary_structure[1:-1] == 0).all()):
            secondary_structure = decode_secondary_structure(input.secondary_structure.cpu(), SecondaryStructureTokenizer())
        if input.sasa is not None and not bool((input.sasa[1:-1] == 0).all()):
            sasa = decode_sasa(input.sasa.cpu(), SASADiscretizingTokenizer())
        if input.sequence is not None and not bool((input.sequence[1:-1] == SEQUENCE_PAD_TOKEN).all()):
            tok = getattr(self.tokenizers, "sequence", None)
            if not hasattr(tok, "decode"):
                tok = EsmSequenceTokenizer()
            sequence = decode_sequence(input.sequence, tok)
        st = input.structure
        if st is not None:
            body = st[1:-1]
            if bool((body == STRUCTURE_PAD_TOKEN).all()) or bool((body == STRUCTURE_MASK_TOKEN).any()):
                st = None
        if st is not None:
            coords, plddt, ptm, pae = decode_structure(st.to(self.device), self.get_structure_decoder())
        elif input.coordinates is not None:
            coords = input.coordinates[1:-1, ...]
        return ESMProtein(sequence=sequence, secondary_structure=secondary_structure, sasa=sasa, coordinates=coords,
                          plddt=plddt, ptm=ptm, pae=pae, potential_sequence_of_concern=input.potential_sequence_of_concern)

    def forward_and_sample(self, input: ESMProteinTensor, sampling_configuration, generator=None):
        """One forward + one sampling step per configured track (reference esm3.py:561-599 with
        utils/generation.py:553-700): integer tracks through the fused kernel, SASA decoded to floats.
        Tracks missing from `input` start fully masked (ESMProteinTensor.empty, api.py:251-278)."""
        import warnings
        from .api import ForwardAndSampleOutput
        from .sampling import MAX_TOPK, default_track_tokenizers, sample_sasa_logits, sample_track
        cfg = sampling_configuration
        if cfg.function is not None:
            raise NotImplementedError("function sampling needs the InterPro tokenizer's data files")
        for name in ("sequence", "structure", "secondary_structure", "sasa"):
            tc = getattr(cfg, name)
            if tc is not None and tc.topk_logprobs > MAX_TOPK:  # validate_sampling_config(on_invalid="warn")
                warnings.warn(f"Sampling track {name} has topk_logprobs={tc.topk_logprobs} greater than MAX_TOPK={MAX_TOPK}.")
        Ln = len(input)
        if Ln <= 0:
            raise ValueError("No input data provided")
        dev = self.device
        toks = default_track_tokenizers()
        tracks = {}
        for name, (bos, eos, _pad, mask) in self._TRACK_IDS.items():
            t = getattr(input, name)
            if t is None:
                t = torch.full((Ln,), mask, dtype=torch.int64)
                t[0], t[-1] = bos, eos
            tracks[name] = t.to(dev)[None]
        coords = None if input.coordinates is None else input.coordinates.to(dev)[None]
        u = lambda x: None if x is None else x.to(dev)[None]
        inp = _BatchedESMProteinTensor(sequence=tracks["sequence"], structure=tracks["structure"],
                                       secondary_structure=tracks["secondary_structure"], sasa=tracks["sasa"],
                                       function=u(input.function), residue_annotations=u(input.residue_annotations),
                                       coordinates=coords)
        lo = self.logits(inp, LogitsConfig(sequence=cfg.sequence is not None, structure=cfg.structure is not None,
                                           secondary_structure=cfg.secondary_structure is not None,
                                           sasa=cfg.sasa is not None, return_embeddings=True))
        out_tokens = {k: v[0].clone() for k, v in tracks.items()}
        meta = {k: {} for k in ("entropy", "prob", "logprob", "top_prob", "topk_logprob", "topk_tokens")}
        for name in ("sequence", "structure", "secondary_structure"):
            tc = getattr(cfg, name)
            if tc is None:
                continue
            tok = toks[name]
            valid = sorted(set(tok.all_token_ids) - set(tok.special_token_ids) - set(tc.invalid_ids))
            m = sample_track(getattr(lo.logits, name).contiguous(), tracks[name], tc, tok.mask_token_id, valid,
                             generator=generator)
            out_tokens[name] = m["sampled_tokens"][0]
            for k in meta:
                if m.get(k) is not None:
                    meta[k][name] = m[k][0]
        if cfg.sasa is not None:
            tok = toks["sasa"]
            valid = sorted(set(tok.all_token_ids) - set(tok.special_token_ids) - set(cfg.sasa.invalid_ids))
            value, ent = sample_sasa_logits(lo.logits.sasa, tracks["sasa"], cfg.sasa, tok.mask_token_id, valid)
            out_tokens["sasa"] = value[0]
            meta["entropy"]["sasa"] = ent[0]
        ftd = lambda dct: ForwardTrackData(**dct) if dct else None
        pt = ESMProteinTensor(sequence=out_tokens["sequence"], structure=out_tokens["structure"],
                              secondary_structure=out_tokens["secondary_structure"], sasa=out_tokens["sasa"],
                              function=input.function, residue_annotations=input.residue_annotations,
                              coordinates=input.coordinates)
        emb = lo.embeddings
        return ForwardAndSampleOutput(
            protein_tensor=pt, entropy=ftd(meta["entropy"]), prob=ftd(meta["prob"]), logprob=ftd(meta["logprob"]),
            top_prob=ftd(meta["top_prob"]), topk_logprob=ftd(meta["topk_logprob"]), topk_tokens=ftd(meta["topk_tokens"]),
            per_residue_embedding=emb[0] if cfg.return_per_residue_embeddings else None,
            mean_embedding=emb[0].mean(dim=0) if cfg.return_mean_embedding else None)
