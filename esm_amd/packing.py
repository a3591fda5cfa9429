"""Host-side batch logic: the reference's three `sequence_id` dialects -> packed varlen segments.

The reference's only masking rule is M[b,i,j] = (sequence_id[b,i] == sequence_id[b,j])
(esm/layers/attention.py:65-72); `None` means no mask (attention.py:73-78).  The HIP attention kernel
works on packed tokens with `cu_seqlens` (the reference's own flash formulation, attention.py:99-101),
so every dialect is converted to contiguous segments here (dialect table):

  None            one segment per batch row, all L positions (pads included - they attend, as in the reference)
  bool, drop=True ESM C semantics (esmc.py:133-164): the True positions of a row form one segment, False
                  positions are dropped and come back as zeros (the reference's flash path; its SDPA path
                  leaves unspecified finite values there)
  int / bool      run-length segments inside each row; when an id re-appears after a different id in the same row
                  (a general equality mask), the row's tokens are stably sorted by id first - `gather` is then that
                  permutation, positions keep the original columns, and callers scatter outputs back

Rotary position = COLUMN index in the [B, L] row, as in the reference's SDPA path (rotary.py:207-209,54-56).
Everything here is numpy on the host and is covered by CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class PackedBatch:
    B: int
    L: int
    n_tokens: int
    cu_seqlens: np.ndarray  # int32 [S+1]
    positions: np.ndarray  # int32 [N]  rotary position (column index)
    gather: np.ndarray | None  # int64 [N] flat index into B*L, None = identity (all tokens kept, in order)
    row_cu: np.ndarray  # int32 [B+1] token range of every batch row in packed order
    seq_id: np.ndarray  # int32 [N] per-token segment label inside its row (geometric-attention soft bias)
    qblk_map: np.ndarray = field(default=None)  # int32 [n_qblk, 2], filled by plan_attention

    @property
    def n_segments(self) -> int:
        return len(self.cu_seqlens) - 1

    @property
    def max_seqlen(self) -> int:
        return int(np.diff(self.cu_seqlens).max()) if self.n_segments else 0

    @property
    def max_row_len(self) -> int:
        return int(np.diff(self.row_cu).max()) if self.B else 0

    @property
    def max_position(self) -> int:
        return int(self.positions.max()) if self.n_tokens else 0


def plan_attention(cu_seqlens: np.ndarray, qblk: int = 128) -> np.ndarray:
    """(segment, first query row) per 128-row query block - same contract as esmk_attn_plan()."""
    out = []
    lens = np.diff(cu_seqlens)
    for s, n in enumerate(lens):
        for q in range(0, int(n), qblk):
            out.append((s, q))
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


def pack_sequence_id(sequence_id, B: int, L: int, drop_false: bool = False) -> PackedBatch:
    """sequence_id: None | np.ndarray [B, L] of bool or integer dtype."""
    cols = np.tile(np.arange(L, dtype=np.int32), B)
    if sequence_id is None:
        cu = (np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)
        pb = PackedBatch(B, L, B * L, cu, cols, None, cu.copy(), np.zeros(B * L, dtype=np.int32))
        pb.qblk_map = plan_attention(pb.cu_seqlens)
        return pb
    sid = np.asarray(sequence_id)
    if sid.shape != (B, L):
        raise ValueError(f"sequence_id shape {sid.shape} != {(B, L)}")
    if sid.dtype == np.bool_ and drop_false:
        keep = sid.reshape(-1)
        gather = np.nonzero(keep)[0].astype(np.int64)
        lens = sid.sum(axis=1).astype(np.int64)
        row_cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        nz = lens > 0
        cu = np.concatenate([[0], np.cumsum(lens[nz])]).astype(np.int32)
        n = int(lens.sum())
        pb = PackedBatch(B, L, n, cu, cols[gather], gather, row_cu, np.zeros(n, dtype=np.int32))
        pb.qblk_map = plan_attention(pb.cu_seqlens)
        return pb
    ids = sid.astype(np.int64)

    def runs(flat_ids):
        """Run boundaries: row starts and positions where the id changes -> (starts, one (row, id) key per run)."""
        m = flat_ids.reshape(B, L)
        change = np.ones((B, L), dtype=bool)
        change[:, 1:] = m[:, 1:] != m[:, :-1]
        st = np.nonzero(change.reshape(-1))[0]
        return st, np.stack([st // L, flat_ids[st]], axis=1)

    flat = ids.reshape(-1)
    starts, key = runs(flat)
    gather = None
    if len(np.unique(key, axis=0)) != len(key):
        # an id re-appears in a non-adjacent run of its row (general equality mask, attention.py:67).  Everything
        # but attention is per-token and attention is permutation-invariant inside a segment once every token
        # carries its own rotary position, so: stable-sort each row by id (equal ids become one contiguous
        # segment), keep the original column as the position, and let the caller scatter the outputs back.
        gather = np.lexsort((np.tile(np.arange(L), B), flat, np.repeat(np.arange(B), L))).astype(np.int64)
        flat = flat[gather]
        starts, key = runs(flat)
        assert len(np.unique(key, axis=0)) == len(key)
    cu = np.concatenate([starts, [B * L]]).astype(np.int32)
    row_cu = (np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)
    # per-token label: equality of labels inside a row must equal equality of ids -> use the id itself (mod 2^31)
    seq_lab = (flat & 0x7FFFFFFF).astype(np.int32)
    pb = PackedBatch(B, L, B * L, cu, cols if gather is None else cols[gather], gather, row_cu, seq_lab)
    pb.qblk_map = plan_attention(pb.cu_seqlens)
    return pb


def pack_lengths(lengths) -> PackedBatch:
    """Already-packed input: sequences back to back, one segment (and one row) each."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if (lengths <= 0).any():
        raise ValueError("sequence lengths must be positive")
    cu = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    n = int(cu[-1])
    pos = np.concatenate([np.arange(k, dtype=np.int32) for k in lengths]) if len(lengths) else np.zeros(0, np.int32)
    pb = PackedBatch(len(lengths), int(lengths.max()) if len(lengths) else 0, n, cu, pos, None, cu.copy(),
                     np.zeros(n, dtype=np.int32))
    pb.qblk_map = plan_attention(pb.cu_seqlens)
    return pb
