"""Strings in, embeddings out: the callers either side of the trunk.

    FASTA / list[str] -> vectorised tokenisation (tokenization.py) -> length-bucketed token-budget packs and
    LPT assignment to ranks (dp.py) -> packed varlen forward on the HIP trunk -> per-sequence mean pooling on the
    device (esmk_segment_mean) or per-residue embeddings -> one gather across ranks.

Replaces the reference's per-call path  `ESMC.encode` -> `_tokenize` -> padded `[B, L]` batch -> `logits`
(esm/models/esmc.py:100-109,175-225) for bulk embedding: no padding tokens are ever computed, the host never
synchronises inside the loop (token uploads come from pinned memory on the compute stream's queue), and the
only collective is the final gather.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist

from .dp import plan
from .engine import DeviceBatch, segment_mean
from .packing import pack_lengths
from .parsing import read_sequences
from .tokenization import EsmSequenceTokenizer


def tokenize_packs(sequences: Sequence[str], tokenizer: EsmSequenceTokenizer, world: int, token_budget: int, d: int, h: int):
    """-> (tokens int64 [sum L], lengths, offsets, packs, per_rank).  Pure host logic (CPU-tested)."""
    tokens, lengths = tokenizer.encode_packed(sequences)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    packs, per_rank, _ = plan(lengths.tolist(), world, token_budget, d, h)
    return tokens, lengths, offsets, packs, per_rank


def gather_pack(tokens: np.ndarray, offsets: np.ndarray, idx: Sequence[int], out: np.ndarray) -> int:
    """Copy the token runs of sequences `idx` back to back into `out`; returns the token count."""
    n = 0
    for i in idx:
        a, b = int(offsets[i]), int(offsets[i + 1])
        out[n:n + b - a] = tokens[a:b]
        n += b - a
    return n


@torch.no_grad()
def embed_sequences(model, sequences: Sequence[str], token_budget: int = 32768, pooling: str = "mean", group=None):
    """Embed protein strings with an `esm_amd.ESMC` model.

    pooling = "mean": returns f32 [n_sequences, d] (mean over each sequence's tokens incl. <cls>/<eos>), on every
    rank.  pooling = "none": returns a list of f32 [L_i, d] per-residue embeddings (local rank's sequences only;
    None for the others) - per-residue output of a large set is not gathered.
    """
    if pooling not in ("mean", "none"):
        raise ValueError("pooling must be 'mean' or 'none'")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    trunk, _head = model._engines()
    dev = model.device
    d = model.d_model
    tokens, lengths, offsets, packs, per_rank = tokenize_packs(sequences, model.tokenizer, world, token_budget, d, trunk.h)
    n_seq = len(sequences)
    pooled = torch.zeros((n_seq, d), dtype=torch.float32, device=dev) if pooling == "mean" else None
    per_res = [None] * n_seq if pooling == "none" else None
    stage = [torch.empty(token_budget, dtype=torch.int64).pin_memory() for _ in range(2)]
    done = [None, None]  # event after the upload that last used the staging buffer
    for k, p in enumerate(per_rank[rank]):
        idx = packs[p]
        buf = stage[k & 1]
        if done[k & 1] is not None:
            done[k & 1].synchronize()  # the upload that used this pinned buffer two packs ago has finished
        n = gather_pack(tokens, offsets, idx, buf.numpy())
        tok_dev = buf[:n].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        done[k & 1] = ev
        batch = DeviceBatch(pack_lengths([int(lengths[i]) for i in idx]), dev)
        _lg, post, _h = model.forward_packed(tok_dev, batch, want_hiddens=False, want_logits=False)
        if pooling == "mean":
            pooled[torch.as_tensor(idx, device=dev)] = segment_mean(post, batch.cu_seqlens)
        else:
            off = 0
            for i in idx:
                per_res[i] = post[off: off + int(lengths[i])]
                off += int(lengths[i])
    if pooling == "none":
        return per_res
    if world > 1:
        dist.all_reduce(pooled, group=group)  # every sequence was written by exactly one rank, zeros elsewhere
    return pooled


def embed_fasta(model, path, token_budget: int = 32768, pooling: str = "mean", group=None):
    """-> (headers, embeddings) for every record of a FASTA file (reference reader: esm/utils/parsing.py:40-58)."""
    entries = list(read_sequences(path))
    return [e.header for e in entries], embed_sequences(model, [e.sequence for e in entries], token_budget, pooling, group)


def save_embeddings(path, headers: Sequence[str], embeddings: torch.Tensor, dtype=torch.float32) -> None:
    """Wire/storage format of a pooled result: one safetensors file holding `embeddings`
    [n, d] plus the FASTA headers (JSON list) in the file metadata - memory-mappable, no pickle."""
    import json
    from safetensors.torch import save_file
    if embeddings.ndim != 2 or embeddings.shape[0] != len(headers):
        raise ValueError("embeddings must be [n_sequences, d] with one header per row")
    save_file({"embeddings": embeddings.detach().to("cpu", dtype).contiguous()}, str(path),
              metadata={"headers": json.dumps(list(headers)), "format": "esm_amd.pooled_embeddings.v1"})


def load_embeddings(path):
    """-> (headers, embeddings) written by save_embeddings."""
    import json
    from safetensors import safe_open
    with safe_open(str(path), framework="pt") as f:
        meta = f.metadata() or {}
        if meta.get("format") != "esm_amd.pooled_embeddings.v1":
            raise ValueError("not an esm_amd pooled-embeddings file")
        return json.loads(meta["headers"]), f.get_tensor("embeddings")
