"""Backbone frames for geometric attention (reference esm/utils/structure/affine3d.py:386-388,494-560).

Only what the trunk consumes is kept: a flat [..., 12] = [rotation (row-major 3x3) | translation] tensor
and the frame-validity mask.  Frames are computed by the HIP kernel behind esmk_frames.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _lib as L


@dataclass(frozen=True)
class Affine3D:
    tensor: torch.Tensor  # [..., 12]

    @property
    def rot(self) -> torch.Tensor:
        return self.tensor[..., :9].unflatten(-1, (3, 3))

    @property
    def trans(self) -> torch.Tensor:
        return self.tensor[..., 9:]

    @property
    def shape(self):
        return self.tensor.shape[:-1]


def build_affine3d_from_coordinates(coords: torch.Tensor):
    """coords [B, L, 3, 3] (N, CA, C) on the GPU -> (Affine3D [B, L], mask bool [B, L])."""
    if not coords.is_cuda:
        raise L.EsmkError("build_affine3d_from_coordinates needs a GPU tensor (no CPU fallback in esm_amd)")
    B, Ln = coords.shape[:2]
    c = coords.to(torch.float32).contiguous()
    aff = torch.empty((B, Ln, 12), dtype=torch.float32, device=c.device)
    mask = torch.empty((B, Ln), dtype=torch.uint8, device=c.device)
    L.check(L.lib().esmk_frames(L.ptr(c), L.ptr(aff), L.ptr(mask), B, Ln, torch.cuda.current_stream().cuda_stream),
            "esmk_frames")
    return Affine3D(aff), mask.bool()
