"""Data-parallel embedding of many proteins across the GPUs of one node.

The path shards by sequence: a protein's forward depends only on its own tokens (block-diagonal attention,
LayerNorm only - no batch statistics), so ranks never talk during the forward.  Weights are replicated.
Host logic (this file, torch.distributed for the one collective):

  1. length-bucket: sort by length, cut into packs of <= token_budget tokens (similar lengths per pack, which
     bounds the 4*L*d attention-term imbalance),
  2. assign packs to ranks, longest-processing-time-first, on the FLOP cost model of SURVEY.md 8(d):
       cost(seq of length L) = L*(8d^2 + 6dh) + 4d*L^2   per layer,
  3. every rank runs its packs through `compute_fn` (the HIP trunk),
  4. ONE all-gather of the per-residue results (padded to the largest rank) + the permutation to undo.

With backend "nccl" the collective is RCCL over xGMI; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist


def sequence_cost(L, d: int, h: int):
    L = np.asarray(L, dtype=np.float64)
    return L * (8.0 * d * d + 6.0 * d * h) + 4.0 * d * L * L


def make_packs(lengths: Sequence[int], token_budget: int) -> list[list[int]]:
    """Indices grouped into packs of <= token_budget tokens, longest sequences first."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if len(lengths) and lengths.max() > token_budget:
        raise ValueError(f"a sequence of {int(lengths.max())} tokens exceeds the token budget {token_budget}")
    order = np.argsort(-lengths, kind="stable")
    packs, cur, used = [], [], 0
    for i in order:
        n = int(lengths[i])
        if cur and used + n > token_budget:
            packs.append(cur)
            cur, used = [], 0
        cur.append(int(i))
        used += n
    if cur:
        packs.append(cur)
    return packs


def assign_packs(pack_costs: Sequence[float], world_size: int) -> list[list[int]]:
    """Longest-processing-time-first assignment of packs to ranks; returns pack indices per rank."""
    loads = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for p in np.argsort(-np.asarray(pack_costs, dtype=np.float64), kind="stable"):
        r = int(np.argmin(loads))
        out[r].append(int(p))
        loads[r] += float(pack_costs[p])
    return out


def plan(lengths: Sequence[int], world_size: int, token_budget: int, d: int, h: int):
    packs = make_packs(lengths, token_budget)
    costs = [float(sequence_cost([lengths[i] for i in p], d, h).sum()) for p in packs]
    per_rank = assign_packs(costs, world_size)
    return packs, per_rank, costs


def run_data_parallel(sequences: Sequence[torch.Tensor], compute_fn: Callable, width: int, d: int, h: int,
                      token_budget: int = 32768, device=None, wire_dtype=torch.bfloat16, group=None):
    """Embed `sequences` (1-D int64 token tensors, identical on every rank) data-parallel.

    compute_fn(tokens_packed [N] on `device`, lengths list[int]) -> [N, width] tensor on `device`.
    Returns a list (original order) of [L_i, width] tensors in `wire_dtype`, available on EVERY rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [int(s.numel()) for s in sequences]
    packs, per_rank, _ = plan(lengths, world, token_budget, d, h)
    device = device or torch.device("cpu")
    # local compute
    chunks = []
    for p in per_rank[rank]:
        idx = packs[p]
        toks = torch.cat([sequences[i] for i in idx]).to(device)
        out = compute_fn(toks, [lengths[i] for i in idx])
        chunks.append(out.to(wire_dtype))
    local = torch.cat(chunks) if chunks else torch.zeros((0, width), dtype=wire_dtype, device=device)
    # token counts per rank are known to everyone from the plan: no size exchange needed
    counts = [sum(lengths[i] for p in per_rank[r] for i in packs[p]) for r in range(world)]
    if world > 1:
        cap = max(counts)
        send = torch.zeros((cap, width), dtype=wire_dtype, device=device)
        send[: local.shape[0]] = local
        recv = torch.empty((world * cap, width), dtype=wire_dtype, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
        per = [recv[r * cap: r * cap + counts[r]] for r in range(world)]
    else:
        per = [local]
    # undo the permutation
    results = [None] * len(sequences)
    for r in range(world):
        off = 0
        for p in per_rank[r]:
            for i in packs[p]:
                results[i] = per[r][off: off + lengths[i]]
                off += lengths[i]
    return results


def esmc_compute_fn(model):
    """compute_fn backed by the HIP trunk: per-residue post-norm embeddings of an ESMC model."""
    from .engine import DeviceBatch
    from .packing import pack_lengths

    def fn(tokens, lengths):
        batch = DeviceBatch(pack_lengths(lengths), tokens.device)
        _logits, post, _ = model.forward_packed(tokens, batch, want_hiddens=False, want_logits=False)
        return post

    return fn
