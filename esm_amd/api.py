"""Minimal re-declaration of the reference's I/O types for the trunk path
(reference esm/sdk/api.py:201-282,368-411; esm/utils/sampling.py:42-106).  Field names and meanings
are the reference's; everything unrelated to the trunk forward (Forge clients, ESMProtein string/PDB
handling, generation configs) is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, fields

import torch


@dataclass
class ESMProtein:
    """Sequence-level view of the reference's ESMProtein (esm/sdk/api.py:37-60): the fields the trunk path uses.
    Structure / function annotation handling (PDB I/O, ProteinChain) is out of scope."""
    sequence: str | None = None
    secondary_structure: str | None = None
    sasa: list | None = None
    coordinates: torch.Tensor | None = None
    plddt: torch.Tensor | None = None
    ptm: torch.Tensor | None = None
    pae: torch.Tensor | None = None
    potential_sequence_of_concern: bool = False

    def __len__(self) -> int:
        if self.sequence is not None:
            return len(self.sequence)
        if self.secondary_structure is not None:
            return len(self.secondary_structure)
        if self.sasa is not None:
            return len(self.sasa)
        if self.coordinates is not None:
            return self.coordinates.size(0)
        raise ValueError("No track to determine length from.")


@dataclass
class ESMProteinTensor:
    sequence: torch.Tensor | None = None
    structure: torch.Tensor | None = None
    secondary_structure: torch.Tensor | None = None
    sasa: torch.Tensor | None = None
    function: torch.Tensor | None = None
    residue_annotations: torch.Tensor | None = None
    coordinates: torch.Tensor | None = None
    potential_sequence_of_concern: bool = False

    def _tensors(self):
        return {f.name: getattr(self, f.name) for f in fields(ESMProteinTensor)
                if isinstance(getattr(self, f.name), torch.Tensor)}

    def __len__(self) -> int:
        lens = {v.size(0) for v in self._tensors().values()}
        if not lens:
            return 0
        if len(lens) != 1:
            raise ValueError(f"Either no tracks or inconsistent length: {lens}")
        return next(iter(lens))

    @property
    def device(self):
        devs = {v.device for v in self._tensors().values()}
        assert len(devs) == 1, devs
        return next(iter(devs))

    def to(self, device_or_dtype):
        for k, v in self._tensors().items():
            setattr(self, k, v.to(device_or_dtype))
        return self


class _BatchedESMProteinTensor(ESMProteinTensor):
    @staticmethod
    def from_protein_tensor(protein: ESMProteinTensor):
        u = lambda x: x.unsqueeze(0) if x is not None else None
        return _BatchedESMProteinTensor(
            sequence=u(protein.sequence), structure=u(protein.structure),
            secondary_structure=u(protein.secondary_structure), sasa=u(protein.sasa), function=u(protein.function),
            residue_annotations=u(protein.residue_annotations), coordinates=u(protein.coordinates))

    def __len__(self) -> int:
        lens = {v.size(1) for v in self._tensors().values()}
        return next(iter(lens)) if lens else 0

    @property
    def batch_size(self) -> int:
        return next(iter({v.size(0) for v in self._tensors().values()}))

    def slice(self, i: int, sequence_len: int | None = None) -> ESMProteinTensor:
        """Row i as an ESMProteinTensor, optionally trimmed to its true length (sampling.py:76-93)."""
        cut = lambda x: None if x is None else (x[i] if sequence_len is None else x[i][:sequence_len])
        return ESMProteinTensor(**{f.name: cut(getattr(self, f.name)) for f in fields(ESMProteinTensor)
                                   if f.name != "potential_sequence_of_concern"})

    def set_slice(self, i: int, slice: ESMProteinTensor):
        """Write `slice` into row i, track by track (sampling.py:95-106)."""
        for f in fields(ESMProteinTensor):
            if f.name == "potential_sequence_of_concern":
                continue
            s, v = getattr(self, f.name), getattr(slice, f.name)
            assert v is None or s is not None, f"Trying to set a slice on None tensor ({f.name})."
            if v is not None:
                s[i, ...] = v


@dataclass
class ESMProteinError(Exception):
    """esm/sdk/api.py:285-288: what batch_generate returns in place of a prompt that could not be decoded."""
    error_code: int  # HTTP convention: 500 = internal error
    error_msg: str


@dataclass
class ForwardTrackData:
    sequence: torch.Tensor | None = None
    structure: torch.Tensor | None = None
    secondary_structure: torch.Tensor | None = None
    sasa: torch.Tensor | None = None
    function: torch.Tensor | None = None


@dataclass
class LogitsConfig:
    sequence: bool = False
    structure: bool = False
    secondary_structure: bool = False
    sasa: bool = False
    function: bool = False
    residue_annotations: bool = False
    return_embeddings: bool = False
    return_hidden_states: bool = False
    return_mean_embedding: bool = False
    return_mean_hidden_states: bool = False
    ith_hidden_layer: int = -1


@dataclass
class LogitsOutput:
    logits: ForwardTrackData | None = None
    embeddings: torch.Tensor | None = None
    mean_embedding: torch.Tensor | None = None
    residue_annotation_logits: torch.Tensor | None = None
    hidden_states: torch.Tensor | None = None
    mean_hidden_state: torch.Tensor | None = None


@dataclass
class ForwardAndSampleOutput(LogitsOutput):
    """esm/sdk/api.py:414-428."""
    protein_tensor: ESMProteinTensor | None = None
    entropy: ForwardTrackData | None = None
    prob: ForwardTrackData | None = None
    logprob: ForwardTrackData | None = None
    top_prob: ForwardTrackData | None = None
    topk_logprob: ForwardTrackData | None = None
    topk_tokens: ForwardTrackData | None = None
    per_residue_embedding: torch.Tensor | None = None
    mean_embedding: torch.Tensor | None = None
