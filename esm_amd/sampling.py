"""Iterative-sampling loop around the trunk.

The reference decodes a masked track by running the forward `num_steps` times and, per prompt and per step,
building half a dozen full-size temporaries in torch (sort + softmax + cumsum for top-p, boolean masks, softmax,
multinomial, log_softmax, Categorical entropy, ...: esm/utils/sampling.py:153-343, esm/utils/generation.py:241-534,
703-822).  Here every step is: one forward, ONE fused kernel per track over all prompts (`esmk_sample_track`),
and a handful of [B, L] tensor ops to choose which positions to commit - logits never leave the device and
nothing is sliced per prompt.

Names and fields follow the reference (`SamplingTrackConfig`, `GenerationConfig`: esm/sdk/api.py:293-343).
Tracks: sequence, structure, secondary_structure, sasa (integer tracks).  `function` needs the InterPro
tokenizer's data files, `coordinates` / `residue_annotations` cannot be sampled iteratively in the reference
either (generation.py:426-431): all three raise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import torch

from . import _lib as L
from .engine import _stream

INTEGER_TRACKS = ("sequence", "structure", "secondary_structure", "sasa")


@dataclass
class SamplingTrackConfig:
    temperature: float = 1.0
    top_p: float = 1.0
    only_sample_masked_tokens: bool = True
    invalid_ids: Sequence[int] = field(default_factory=list)
    topk_logprobs: int = 0


@dataclass
class GenerationConfig:
    track: str = ""
    invalid_ids: Sequence[int] = field(default_factory=list)
    schedule: str = "cosine"       # how many tokens stay masked after each step
    strategy: str = "random"       # which ones are committed: "random" | "entropy" (lowest entropy first)
    num_steps: int = 20
    temperature: float = 1.0
    temperature_annealing: bool = True
    top_p: float = 1.0
    condition_on_coordinates_only: bool = True

    def __post_init__(self):
        if self.schedule not in NOISE_SCHEDULES:
            raise ValueError(f"schedule must be one of {sorted(NOISE_SCHEDULES)}")
        if self.strategy not in ("random", "entropy"):
            raise ValueError("strategy must be 'random' or 'entropy'")

    def use_entropy_based_unmasking_strategy(self):
        self.schedule, self.strategy, self.temperature_annealing = "cosine", "entropy", False

    def use_generative_unmasking_strategy(self):
        self.schedule, self.strategy, self.temperature_annealing = "cosine", "random", True


@dataclass
class SamplingConfig:
    """esm/sdk/api.py:346-365: one SamplingTrackConfig per track to sample (None = keep the input track)."""
    sequence: SamplingTrackConfig | None = None
    structure: SamplingTrackConfig | None = None
    secondary_structure: SamplingTrackConfig | None = None
    sasa: SamplingTrackConfig | None = None
    function: SamplingTrackConfig | None = None
    return_per_residue_embeddings: bool = False
    return_mean_embedding: bool = False


MAX_TOPK = 32  # esm/utils/constants/api.py:1-5
SASA_DISCRETIZATION_BOUNDARIES = [0.8, 4.0, 9.6, 16.4, 24.5, 32.9, 42.0, 51.5, 61.2, 70.9, 81.6, 93.3, 107.2, 125.4,
                                  151.4]  # esm/utils/constants/esm3.py:74-90


# esm/utils/noise_schedules.py:6-35 - fraction of the track still masked at progress t (a float32 tensor, as in
# the reference, so that the truncation in step_unmask_count sees the same rounding)
NOISE_SCHEDULES = {
    "cosine": lambda t: torch.cos(t * math.pi * 0.5),
    "linear": lambda t: 1 - t,
    "square_root_schedule": lambda t: 1 - torch.sqrt(t),
    "cubic": lambda t: 1 - t ** 3,
    "square": lambda t: 1 - t ** 2,
}


def annealed_temperature(step: int, num_steps: int, initial_temperature: float) -> float:
    """generation.py:350-352."""
    step_ratio = step / max(1, (num_steps - 1))
    return max(initial_temperature - step_ratio, 0.001) ** 2


def get_sampling_mask(tokens: torch.Tensor, config: SamplingTrackConfig, mask_idx: int) -> torch.Tensor:
    """Positions the sampler may overwrite, bool [B, L] (reference sampling.py:321-342): `esmk_sampling_mask`."""
    import ctypes as C
    assert tokens.is_cuda and tokens.ndim == 2
    B, Ln = tokens.shape
    tok = tokens.to(torch.int64).contiguous()
    ids = sorted(set(int(i) for i in config.invalid_ids))
    if len(ids) > 64:
        raise ValueError("at most 64 special ids")
    host = (C.c_int64 * max(len(ids), 1))(*ids)
    out = torch.empty((B, Ln), dtype=torch.uint8, device=tokens.device)
    L.check(L.lib().esmk_sampling_mask(L.ptr(tok), B, Ln, host, len(ids), int(mask_idx),
                                       1 if config.only_sample_masked_tokens else 0, L.ptr(out), _stream()), "esmk_sampling_mask")
    return out.bool()


def sample_track(logits: torch.Tensor, tokens: torch.Tensor, config: SamplingTrackConfig, mask_idx: int,
                 valid_ids: Sequence[int], noise: torch.Tensor | None = None, generator=None,
                 mask_logits_of_invalid_ids: bool = True) -> dict:
    """`_sample_track` + `_compute_track_metadata` (generation.py:703-733,786-822) as one kernel.

    logits f32 [B, L, V] on the GPU, contiguous: MODIFIED IN PLACE like the reference's (top-p / invalid ids).
    noise: optional f32 [B*L, V] of Exp(1) draws (tests pass the ones torch.multinomial would draw)."""
    if len(valid_ids) == 0:
        raise ValueError("Can not sample logits if there are no valid ids to sample from.")
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.ndim == 3
    B, Ln, V = logits.shape
    assert tokens.shape == (B, Ln)
    dev = logits.device
    valid = None
    if mask_logits_of_invalid_ids:
        valid = torch.zeros(V, dtype=torch.uint8, device=dev)
        valid[torch.as_tensor(sorted(valid_ids), device=dev)] = 1
    smask = get_sampling_mask(tokens, config, mask_idx)
    temperature = float(config.temperature)
    if temperature > 0 and noise is None:
        noise = torch.empty((B * Ln, V), dtype=torch.float32, device=dev).exponential_(1, generator=generator)
    R = B * Ln
    tok = tokens.reshape(-1).to(torch.int64).contiguous()
    sm8 = smask.reshape(-1).to(torch.uint8).contiguous()
    sampled = torch.empty(R, dtype=torch.int64, device=dev)
    ent, lp, tp = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(3))
    L.check(L.lib().esmk_sample_track(L.ptr(logits), R, V, L.ptr(valid), temperature, float(config.top_p), L.ptr(noise),
                                      L.ptr(tok), L.ptr(sm8), L.ptr(sampled), L.ptr(ent), L.ptr(lp), L.ptr(tp), _stream()),
            "esmk_sample_track")
    out = {"sampled_tokens": sampled.view(B, Ln), "entropy": ent.view(B, Ln), "logprob": lp.view(B, Ln),
           "prob": torch.exp(lp).view(B, Ln), "top_prob": tp.view(B, Ln), "sampling_mask": smask,
           "topk_logprob": None, "topk_tokens": None}
    if config.topk_logprobs > 0:
        out["topk_logprob"], out["topk_tokens"] = torch.topk(torch.log_softmax(logits, -1), config.topk_logprobs, dim=-1)
    return out


_SASA_MID: dict = {}


def sample_sasa_logits(logits: torch.Tensor, tokens: torch.Tensor, config: SamplingTrackConfig, mask_idx: int,
                       valid_ids: Sequence[int], mask_logits_of_invalid_ids: bool = True):
    """SASA comes back as a float per residue (reference sampling.py:255-290; entropy generation.py:631-639):
    `esmk_sample_sasa` on the [B, L, 19] logits, which are edited in place like the reference's.
    Returns (sasa_value [B, L], entropy [B, L])."""
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.ndim == 3
    B, Ln, V = logits.shape
    dev = logits.device
    mid = _SASA_MID.get(dev)
    if mid is None:  # mid-points of the value bins [0, b_0), [b_0, b_1), ...
        edges = torch.tensor([0.0] + list(SASA_DISCRETIZATION_BOUNDARIES), dtype=torch.float32)
        mid = _SASA_MID[dev] = ((edges[1:] + edges[:-1]) * 0.5).to(dev)
    assert mid.numel() == V - 4
    valid = None
    if mask_logits_of_invalid_ids:
        valid = torch.zeros(V, dtype=torch.uint8)
        valid[list(valid_ids)] = 1
        valid = valid.to(dev)
    smask = get_sampling_mask(tokens, config, mask_idx).to(torch.uint8)
    value = torch.empty((B, Ln), dtype=torch.float32, device=dev)
    entropy = torch.empty((B, Ln), dtype=torch.float32, device=dev)
    L.check(L.lib().esmk_sample_sasa(L.ptr(logits), B * Ln, V, Ln, L.ptr(valid), L.ptr(mid), L.ptr(smask), L.ptr(value),
                                     L.ptr(entropy), _stream()), "esmk_sample_sasa")
    return value, entropy


def step_unmask_count(step: int, num_steps: int, schedule: str, total_to_sample: torch.Tensor,
                      still_masked: torch.Tensor) -> torch.Tensor:
    """How many positions each prompt commits at `step` (generation.py:287-300), batched over prompts."""
    frac = NOISE_SCHEDULES[schedule](torch.tensor((step + 1) / num_steps))  # float32 scalar tensor
    after = (frac * total_to_sample.cpu() + 0.1).int()  # int64 * float32 -> float32, truncated like the reference
    return still_masked.cpu().int() - after


def choose_positions(candidates: torch.Tensor, num_to_sample: torch.Tensor, strategy: str, entropy: torch.Tensor | None,
                     generator=None) -> torch.Tensor:
    """Pick `num_to_sample[b]` of the candidate positions of every row: lowest entropy first, or at random
    (generation.py:302-328).  candidates bool [B, L] -> bool [B, L]."""
    B, Ln = candidates.shape
    if strategy == "entropy":
        key = entropy.masked_fill(~candidates, torch.finfo(entropy.dtype).max)
    else:
        key = torch.rand((B, Ln), device=candidates.device, generator=generator).masked_fill(~candidates, 2.0)
    order = torch.argsort(key, dim=-1, stable=True)
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(Ln, device=candidates.device).expand(B, Ln))
    return candidates & (rank < num_to_sample.to(candidates.device)[:, None].to(rank.dtype))


class _Tok:
    """The token-id facts the loop needs from a track tokenizer (reference tokenizers: mask / all / special ids)."""

    def __init__(self, mask_token_id, n_tokens, special_token_ids):
        self.mask_token_id, self.all_token_ids, self.special_token_ids = mask_token_id, list(range(n_tokens)), list(special_token_ids)


def default_track_tokenizers():
    from .tokenization import EsmSequenceTokenizer, SASADiscretizingTokenizer, SecondaryStructureTokenizer
    return {
        "sequence": EsmSequenceTokenizer(),
        "structure": _Tok(4096, 4101, [4096, 4097, 4098, 4099, 4100]),   # structure_tokenizer.py:10-17
        "secondary_structure": SecondaryStructureTokenizer(),              # 11 ids, <pad> doubles as mask
        "sasa": SASADiscretizingTokenizer(),                               # 19 ids, same convention
    }


@torch.no_grad()
def iterative_sampling(logits_fn, tokens: torch.Tensor, lengths: Sequence[int], config: GenerationConfig, tokenizer,
                       generator=None):
    """Decode the masked positions of `config.track` for a padded batch of prompts.

    logits_fn(track_tokens [B, L]) -> f32 [B, L, V] logits of that track on the GPU (one trunk forward).
    tokens int64 [B, L] (right-padded), lengths[b] = true length of prompt b incl. BOS/EOS.
    Mirrors iterative_sampling_tokens (generation.py:355-534) with every prompt sharing `config`; returns the
    completed tokens [B, L].
    """
    if config.track in ("coordinates", "residue_annotations"):
        raise ValueError(f"Iterative sampling {config.track} is not supported.")  # generation.py:426-431
    if config.track not in INTEGER_TRACKS:
        raise NotImplementedError(f"track {config.track!r}: only {INTEGER_TRACKS} are sampled here")
    dev = tokens.device
    B, Ln = tokens.shape
    cur = tokens.clone()
    mask_idx = tokenizer.mask_token_id
    lens = torch.as_tensor(list(lengths), device=dev)
    col = torch.arange(Ln, device=dev)[None, :]
    inside = (col >= 1) & (col < (lens - 1)[:, None])  # not BOS, not EOS / padding (generation.py:268-273)
    total = ((cur == mask_idx) & inside).sum(-1)       # positions to decode per prompt
    if int(total.max()) == 0:
        raise ValueError(f"Cannot sample {config.track} when input has no masks.")
    num_steps = max(1, min(config.num_steps, int(total.max())))
    valid_ids = sorted(set(tokenizer.all_token_ids) - set(tokenizer.special_token_ids) - set(config.invalid_ids))
    for t in range(num_steps):
        logits = logits_fn(cur)
        temperature = annealed_temperature(t, num_steps, config.temperature) if config.temperature_annealing else config.temperature
        tc = SamplingTrackConfig(temperature=temperature, top_p=config.top_p, invalid_ids=config.invalid_ids)
        meta = sample_track(logits.contiguous(), cur, tc, mask_idx, valid_ids, generator=generator)
        cand = (cur == mask_idx) & inside
        n = step_unmask_count(t, num_steps, config.schedule, total.cpu(), cand.sum(-1).cpu())
        where = choose_positions(cand, n.clamp(min=0), config.strategy, meta["entropy"], generator=generator)
        cur = torch.where(where, meta["sampled_tokens"], cur)
    return cur
