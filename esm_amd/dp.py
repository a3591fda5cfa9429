"""Data-parallel embedding of many proteins across the GPUs of one node.

The path shards by sequence: a protein's forward depends only on its own tokens (block-diagonal attention,
LayerNorm only - no batch statistics), so ranks never talk during the forward.  Weights are replicated.
Host logic (this file, torch.distributed for the one collective):

  1. length-bucket: sort by length, cut into packs of <= token_budget tokens (similar lengths per pack, which
     bounds the 4*L*d attention-term imbalance),
  2. assign packs to ranks, longest-processing-time-first, on the FLOP cost model of SURVEY.md 8(d):
       cost(seq of length L) = L*(8d^2 + 6dh) + 4d*L^2   per layer,
  3. every rank runs its packs through `compute_fn` (the HIP trunk), one pack per round,
  4. the per-residue results of a round are gathered (to all ranks or to rank 0) while the next round computes;
     two buffer pairs alternate, the permutation is undone as results are handed over.

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
    """Indices grouped into packs of <= token_budget tokens: best-fit decreasing bin packing (longest sequences first,
    each into the fullest pack that still takes it; a pack's indices stay in descending length order).  Packs come out
    full to within the shortest sequence, which matters twice: a pack's GEMMs run whole rounds of tiles whether the
    last round is full or not, and fewer packs mean fewer rounds per rank (the bench's 4096 sequences: 72 packs at
    99.99 % fill instead of the 73 at 97.8 % that closing a pack at the first misfit gave).  O(n log n) searches on a
    sorted list of remaining capacities."""
    import bisect
    lengths = np.asarray(lengths, dtype=np.int64)
    if len(lengths) and lengths.max() > token_budget:
        raise ValueError(f"a sequence of {int(lengths.max())} tokens exceeds the token budget {token_budget}")
    if len(lengths) == 0:
        return []
    shortest = int(lengths.min())
    order = np.argsort(-lengths, kind="stable")
    packs: list[list[int]] = []
    room: list[tuple[int, int]] = []  # (remaining tokens, pack index) of the packs that can still take something, sorted
    for i in order:
        n = int(lengths[i])
        k = bisect.bisect_left(room, (n, -1))
        if k == len(room):
            idx, left = len(packs), token_budget - n
            packs.append([int(i)])
        else:
            left, idx = room.pop(k)
            left -= n
            packs[idx].append(int(i))
        if left >= shortest:
            bisect.insort(room, (left, idx))
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
                      token_budget: int = 32768, device=None, wire_dtype=torch.bfloat16, group=None,
                      gather: str = "all", consume: Callable | None = None, stats: dict | None = None):
    """Embed `sequences` (1-D int64 token tensors, identical on every rank) data-parallel, streaming.

    compute_fn(tokens_packed [N] on `device`, lengths list[int]) -> [N, width] tensor on `device`.

    The packs of a rank run one after the other; round k = the k-th pack of every rank.  The results of round k
    travel (one collective per round, sizes known to everyone from the plan - no size exchange) while round k+1
    computes: on a GPU the collective runs on a side stream, and two send/receive buffer pairs alternate, so at most
    TWO rounds of results are resident besides what the caller keeps.

    gather  "all"   every rank receives every result (all_gather_into_tensor, RCCL with backend nccl)
            "root"  only rank 0 receives them (dist.gather); other ranks return None for every sequence
            "none"  no collective: a rank keeps the results of its own sequences (None for the others)
    consume optional callback consume(indices list[int], tensors list[[L_i, width]]) called once per round and source
            rank with results that are only valid during the call (views into a buffer that is reused two rounds
            later): pool, write out or copy them there.  With a callback nothing is retained and the function
            returns None - this is the form for large jobs (config C4: 4096 sequences would otherwise be several
            GB per rank).
    stats   optional dict, filled with {"rounds", "peak_resident_rounds", "buffer_bytes", "collectives"}.
    Without `consume`: returns a list (original order) of [L_i, width] tensors in `wire_dtype` (copies).
    """
    if gather not in ("all", "root", "none"):
        raise ValueError("gather must be 'all', 'root' or 'none'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = [int(s.numel()) for s in sequences]
    packs, per_rank, _ = plan(lengths, world, token_budget, d, h)
    device = torch.device(device) if device is not None else torch.device("cpu")
    on_gpu = device.type == "cuda"
    n_rounds = max((len(p) for p in per_rank), default=0)
    # tokens of rank r in round k (0 when it has run out of packs): known to every rank from the plan
    count = lambda r, k: sum(lengths[i] for i in packs[per_rank[r][k]]) if k < len(per_rank[r]) else 0
    cap = max((count(r, k) for r in range(world) for k in range(n_rounds)), default=0)
    collective = world > 1 and gather != "none"
    results = None if consume is not None else [None] * len(sequences)
    st = {"rounds": n_rounds, "peak_resident_rounds": 0, "buffer_bytes": 0, "collectives": 0}

    send = [torch.empty((cap, width), dtype=wire_dtype, device=device) for _ in range(2)]
    recv = None
    if collective and (gather == "all" or rank == 0):
        recv = [torch.empty((world * cap, width), dtype=wire_dtype, device=device) for _ in range(2)]
    st["buffer_bytes"] = sum(t.numel() * t.element_size() for t in send + (recv or []))
    side = torch.cuda.Stream(device=device) if (on_gpu and collective) else None
    done = [None, None]   # per buffer pair: event (GPU) marking the end of its collective
    resident = 0

    def deliver(k):
        """Hand round k's results to the caller (consume) or copy them into the retained list."""
        nonlocal resident
        slot = k & 1
        if side is not None:
            torch.cuda.current_stream(device).wait_event(done[slot])
        if collective:
            srcs = range(world) if recv is not None else ()
            view = lambda r: recv[slot][r * cap: r * cap + count(r, k)]
        else:
            srcs = (rank,)
            view = lambda r: send[slot][: count(r, k)]
        for r in srcs:
            if k >= len(per_rank[r]):
                continue
            idx = packs[per_rank[r][k]]
            buf, off, parts = view(r), 0, []
            for i in idx:
                parts.append(buf[off: off + lengths[i]])
                off += lengths[i]
            if consume is not None:
                consume(idx, parts)
            else:
                for i, t in zip(idx, parts):
                    results[i] = t.clone()
        resident -= 1

    for k in range(n_rounds):
        slot = k & 1
        # buffers of this slot were last used by round k-2, which was handed over in iteration k-1
        n = count(rank, k)
        if n:
            idx = packs[per_rank[rank][k]]
            toks = torch.cat([sequences[i] for i in idx]).to(device, non_blocking=True)
            out = compute_fn(toks, [lengths[i] for i in idx])
            send[slot][:n].copy_(out)   # cast to the wire dtype; the pack's own output is dropped right here
            del out
        resident += 1
        st["peak_resident_rounds"] = max(st["peak_resident_rounds"], resident)
        if collective:
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(device))
                ctx = torch.cuda.stream(side)
            else:
                import contextlib
                ctx = contextlib.nullcontext()
            with ctx:
                if gather == "all":
                    dist.all_gather_into_tensor(recv[slot], send[slot], group=group)
                else:
                    parts = list(recv[slot].view(world, cap, width).unbind(0)) if rank == 0 else None
                    # dist.gather's dst is a GLOBAL rank; `rank` above is the rank inside `group`
                    root = dist.get_global_rank(group, 0) if group is not None else 0
                    dist.gather(send[slot], parts, dst=root, group=group)
                if side is not None:
                    done[slot] = torch.cuda.Event()
                    done[slot].record(side)
            st["collectives"] += 1
        # round k's collective is now in flight (GPU: on the side stream) while we hand over round k-1; the next
        # iteration's compute is enqueued behind that hand-over and runs beside round k's collective
        if k >= 1:
            deliver(k - 1)
    if n_rounds:
        deliver(n_rounds - 1)
    if side is not None:
        torch.cuda.current_stream(device).wait_stream(side)
    if stats is not None:
        stats.update(st)
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
