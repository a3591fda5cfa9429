"""Sequence tokenizer and token packing for the input side of the trunk.

`EsmSequenceTokenizer` reproduces the ids of the reference's character-level tokenizer
(esm/tokenization/sequence_tokenizer.py:10-66: BPE without merges over SEQUENCE_VOCAB,
esm/utils/constants/esm3.py:51-58; `<cls> $A <eos>` template) without the `tokenizers` machinery: plain
ASCII sequences go through one 256-entry lookup table (vectorised over a whole batch), anything containing
`<...>` special-token strings or non-ASCII characters takes a per-character path with the same result.
`tokenize_sequence` adds the reference's `_` -> `<mask>` convention (esm/utils/encoding.py:35-45).
"""
from __future__ import annotations

import re
from typing import Sequence

import numpy as np
import torch

SEQUENCE_VOCAB = [
    "<cls>", "<pad>", "<eos>", "<unk>",
    "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K",
    "Q", "N", "F", "Y", "M", "H", "W", "C", "X", "B", "U", "Z",
    "O", ".", "-", "|",
    "<mask>",
]  # ids are positions: esm/utils/constants/esm3.py:51-58
MASK_STR_SHORT = "_"
_SPECIAL_RE = re.compile(r"(<cls>|<pad>|<eos>|<unk>|<mask>)")


class EsmSequenceTokenizer:
    """Same ids and special-token attributes as the reference tokenizer."""

    cls_token, pad_token, eos_token, unk_token, mask_token, chain_break_token = (
        "<cls>", "<pad>", "<eos>", "<unk>", "<mask>", "|")

    def __init__(self):
        self.all_tokens = list(SEQUENCE_VOCAB)
        self.token_to_id = {t: i for i, t in enumerate(self.all_tokens)}
        self.cls_token_id = self.bos_token_id = self.token_to_id["<cls>"]
        self.pad_token_id = self.token_to_id["<pad>"]
        self.eos_token_id = self.token_to_id["<eos>"]
        self.unk_token_id = self.token_to_id["<unk>"]
        self.mask_token_id = self.token_to_id["<mask>"]
        self.chain_break_token_id = self.token_to_id["|"]
        lut = np.full(256, self.unk_token_id, dtype=np.int64)
        for t, i in self.token_to_id.items():
            if len(t) == 1:
                lut[ord(t)] = i
        self._lut = lut
        self._lut_mask = lut.copy()  # tokenize_sequence: "_" means <mask>
        self._lut_mask[ord(MASK_STR_SHORT)] = self.mask_token_id

    @property
    def vocab_size(self) -> int:
        return len(self.all_tokens)

    @property
    def bos_token(self):
        return self.cls_token

    @property
    def all_token_ids(self):
        return list(range(self.vocab_size))

    @property
    def special_token_ids(self):  # HF all_special_ids of the reference tokenizer: eos, unk, pad, cls, mask, "|"
        return [self.eos_token_id, self.unk_token_id, self.pad_token_id, self.cls_token_id, self.mask_token_id,
                self.chain_break_token_id]

    # ---- single sequences ---------------------------------------------------------------------
    def _body(self, sequence: str, lut: np.ndarray) -> np.ndarray:
        if "<" not in sequence and sequence.isascii():
            return lut[np.frombuffer(sequence.encode("ascii"), dtype=np.uint8)]
        ids = []
        for piece in _SPECIAL_RE.split(sequence):
            if piece in self.token_to_id and len(piece) > 1:
                ids.append(self.token_to_id[piece])
            else:  # one id per CHARACTER (a multi-byte character is one <unk>)
                ids.extend(int(lut[ord(ch)]) if ord(ch) < 256 else self.unk_token_id for ch in piece)
        return np.asarray(ids, dtype=np.int64)

    def encode(self, sequence: str, add_special_tokens: bool = True) -> list[int]:
        body = self._body(sequence, self._lut)
        if add_special_tokens:
            return [self.cls_token_id, *body.tolist(), self.eos_token_id]
        return body.tolist()

    def decode(self, ids) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        return " ".join(self.all_tokens[int(i)] for i in ids)

    # ---- batches ------------------------------------------------------------------------------
    def encode_packed(self, sequences: Sequence[str], add_special_tokens: bool = True, mask_underscore: bool = True):
        """All sequences back to back -> (tokens int64 [sum L_i], lengths int64 [n]).  One table lookup over the
        concatenated bytes when every sequence is plain ASCII (the common case)."""
        lut = self._lut_mask if mask_underscore else self._lut
        n = len(sequences)
        extra = 2 if add_special_tokens else 0
        if all(("<" not in s and s.isascii()) for s in sequences):
            body_len = np.fromiter((len(s) for s in sequences), dtype=np.int64, count=n)
            body = lut[np.frombuffer("".join(sequences).encode("ascii"), dtype=np.uint8)]
        else:
            bodies = [self._body(s.replace(MASK_STR_SHORT, self.mask_token) if mask_underscore else s, self._lut)
                      for s in sequences]
            body_len = np.fromiter((len(b) for b in bodies), dtype=np.int64, count=n)
            body = np.concatenate(bodies) if bodies else np.zeros(0, np.int64)
        lengths = body_len + extra
        if not add_special_tokens:
            return body, lengths
        out = np.empty(int(lengths.sum()), dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
        is_body = np.ones(out.shape[0], dtype=bool)
        is_body[starts] = False
        is_body[starts + lengths - 1] = False
        out[starts] = self.cls_token_id
        out[starts + lengths - 1] = self.eos_token_id
        out[is_body] = body
        return out, lengths


def tokenize_sequence(sequence: str, sequence_tokenizer: EsmSequenceTokenizer, add_special_tokens: bool = True) -> torch.Tensor:
    """reference esm/utils/encoding.py:35-45."""
    sequence = sequence.replace(MASK_STR_SHORT, sequence_tokenizer.mask_token)
    return torch.tensor(sequence_tokenizer.encode(sequence, add_special_tokens=add_special_tokens), dtype=torch.int64)


def decode_sequence(sequence_tokens, sequence_tokenizer: EsmSequenceTokenizer) -> str:
    """reference esm/utils/decoding.py:125-135."""
    s = sequence_tokenizer.decode(sequence_tokens).replace(" ", "")
    s = s.replace(sequence_tokenizer.mask_token, MASK_STR_SHORT)
    return s.replace(sequence_tokenizer.cls_token, "").replace(sequence_tokenizer.eos_token, "")


# ---- secondary-structure and SASA tracks (no data files needed: esm/tokenization/ss_tokenizer.py,
#      sasa_tokenizer.py; esm/utils/encoding.py:100-135, decoding.py:172-203) --------------------------------------
SSE_8CLASS_VOCAB = "GHITEBSC"   # esm/utils/constants/esm3.py
SSE_3CLASS_VOCAB = "HEC"
SASA_DISCRETIZATION_BOUNDARIES = [0.8, 4.0, 9.6, 16.4, 24.5, 32.9, 42.0, 51.5, 61.2, 70.9, 81.6, 93.3, 107.2, 125.4, 151.4]


class _PadIsEverything:
    """Both tokenizers use <pad> (id 0) as mask, BOS, EOS, pad and chain break (ss_tokenizer.py:79-117)."""
    special_tokens = ["<pad>", "<motif>", "<unk>"]
    mask_token = bos_token = eos_token = pad_token = chain_break_token = "<pad>"
    mask_token_id = bos_token_id = eos_token_id = pad_token_id = chain_break_token_id = 0

    @property
    def vocab_to_index(self) -> dict:
        return {w: i for i, w in enumerate(self.vocab)}

    @property
    def all_token_ids(self):
        return list(range(len(self.vocab)))

    @property
    def special_token_ids(self):
        return [0, 1, 2]

    def get_special_tokens_mask(self, tokens: torch.Tensor) -> torch.Tensor:
        return tokens < len(self.special_tokens)


class SecondaryStructureTokenizer(_PadIsEverything):
    def __init__(self, kind: str = "ss8"):
        assert kind in ("ss8", "ss3")
        self.kind = kind
        self.vocab = [*self.special_tokens, *(SSE_8CLASS_VOCAB if kind == "ss8" else SSE_3CLASS_VOCAB)]

    def encode(self, sequence, add_special_tokens: bool = True) -> torch.Tensor:
        v = self.vocab_to_index
        ids = [v[ch] for ch in sequence]  # KeyError on an unknown symbol, like the reference
        if add_special_tokens:
            ids = [0, *ids, 0]
        return torch.tensor(ids, dtype=torch.int64)

    def decode(self, encoded: torch.Tensor) -> str:
        return "".join(self.vocab[int(i)] for i in encoded)


class SASADiscretizingTokenizer(_PadIsEverything):
    def __init__(self, boundaries=SASA_DISCRETIZATION_BOUNDARIES):
        self._boundaries = sorted(boundaries)
        b = ["0"] + [str(x) for x in self._boundaries] + ["inf"]
        self.vocab = self.special_tokens + [f"<{lo}-{hi}>" for lo, hi in zip(b[:-1], b[1:])]
        e = [0] + self._boundaries + [self._boundaries[-1] * 2]
        self.midpoints_tensor = torch.Tensor([float("nan")] * 3 + [(float(hi) + float(lo)) / 2 for lo, hi in zip(e[:-1], e[1:])])

    def midpoints(self) -> list:
        return self.midpoints_tensor.tolist()

    def encode(self, values, add_special_tokens: bool = True) -> torch.Tensor:
        v, bnd, ids = self.vocab_to_index, torch.tensor(self._boundaries), []
        for value in values:
            if isinstance(value, (float, int)):
                ids.append(3 + int(torch.bucketize(torch.tensor(float(value)), bnd)))
            elif isinstance(value, str):
                ids.append(v[value])
            else:
                raise TypeError(value)
        if add_special_tokens:
            ids = [0, *ids, 0]
        return torch.tensor(ids, dtype=torch.int64)

    def decode_float(self, encoded: torch.Tensor) -> list:
        # special tokens decode to NaN (the reference assigns None into a float array, which stores NaN)
        return self.midpoints_tensor[encoded.cpu()].tolist()

    def decode_list(self, encoded: torch.Tensor) -> list:
        return [self.vocab[int(i)] for i in encoded]

    def decode(self, encoded: torch.Tensor) -> str:
        return ",".join(self.decode_list(encoded))


def tokenize_secondary_structure(secondary_structure, tokenizer: SecondaryStructureTokenizer,
                                 add_special_tokens: bool = True) -> torch.Tensor:
    """encoding.py:100-123: one symbol per residue, '_' (or the tokenizer's own mask string) = masked."""
    if isinstance(secondary_structure, str):
        secondary_structure = secondary_structure.replace(tokenizer.mask_token, MASK_STR_SHORT)
    chars = [tokenizer.mask_token if ch == MASK_STR_SHORT else ch for ch in secondary_structure]
    return tokenizer.encode(chars, add_special_tokens=add_special_tokens)


def tokenize_sasa(sasa, tokenizer: SASADiscretizingTokenizer, add_special_tokens: bool = True) -> torch.Tensor:
    """encoding.py:126-135: None = masked."""
    return tokenizer.encode([tokenizer.mask_token if v is None else v for v in sasa], add_special_tokens=add_special_tokens)


def decode_secondary_structure(tokens: torch.Tensor, tokenizer: SecondaryStructureTokenizer) -> str:
    return tokenizer.decode(tokens[1:-1])  # decoding.py:172-178


def decode_sasa(tokens: torch.Tensor, tokenizer: SASADiscretizingTokenizer) -> list:
    """decoding.py:181-203: integer tokens -> bin mid-points (NaN for special tokens); float values (what
    forward_and_sample returns for this track) pass through with NaN -> None."""
    if tokens[0] != 0:
        raise ValueError("SASA does not start with 0 corresponding to BOS token")
    if tokens[-1] != 0:
        raise ValueError("SASA does not end with 0 corresponding to EOS token")
    body = tokens[1:-1]
    if body.dtype in (torch.int8, torch.int16, torch.int32, torch.int64):
        return tokenizer.decode_float(body)
    return [None if x != x else x for x in body.tolist()]


def stack_variable_length_tensors(sequences: Sequence[torch.Tensor], constant_value=0, dtype=None) -> torch.Tensor:
    """Right-pad to a common shape (reference esm/utils/misc.py:177-206)."""
    shape = [len(sequences)] + np.max([list(s.shape) for s in sequences], 0).tolist()
    dtype = dtype or sequences[0].dtype
    out = torch.full(shape, constant_value, dtype=dtype, device=sequences[0].device)
    for row, s in zip(out, sequences):
        row[tuple(slice(n) for n in s.shape)] = s
    return out
