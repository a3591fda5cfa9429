"""The CPU oracle against the IMPORTED reference on inputs that are not stored anywhere (fresh seeds, other shapes than
the fixtures): runs only where /root/reference exists (the build container; skipped on the GPU box, which holds the
fixtures instead).  Checker against checker - no product code involved."""
import pytest
import torch

from esm_amd import synth
from oracle import ref_import, trunk_ref

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")


def _close(a, b, tol=2e-5):
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("seed", [101, 202])
def test_esmc_forward_all_sequence_id_dialects(seed):
    d, H, NL = 192, 3, 4
    w = synth.esmc_weights(d, NL, seed=seed)
    ref = ref_import.build_esmc(d, H, NL, w)
    tok = synth.synth_tokens([37, 50, 9], pad_to=50, seed=seed)
    g = torch.Generator().manual_seed(seed)
    packed = torch.randint(0, 3, (3, 50), generator=g).sort(dim=1).values          # contiguous (bin-packed) ids
    scattered = torch.randint(0, 3, (3, 50), generator=g)                            # general equality mask
    for sid in (None, tok != 1, packed, scattered):
        with torch.no_grad():
            r = ref(tok, sequence_id=sid)
        o = trunk_ref.esmc_forward(tok, w, H, NL, sequence_id=sid)
        valid = tok != 1 if sid is None or sid.dtype == torch.bool else torch.ones_like(tok, dtype=torch.bool)
        _close(o["embeddings"][valid], r.embeddings[valid])
        _close(o["sequence_logits"][valid], r.sequence_logits[valid])
        _close(o["hidden_states"][:, valid], r.hidden_states[:, valid])


def test_esm3_forward_with_chains_and_missing_coordinates():
    d, H, VH, NL, B, L = 128, 2, 8, 3, 2, 31
    w = synth.esm3_weights(d, NL, VH, seed=303)
    ref = ref_import.build_esm3(d, H, VH, NL, w)
    g = torch.Generator().manual_seed(303)
    kw = dict(sequence_tokens=synth.synth_tokens([31, 20], pad_to=L, seed=5),
              structure_tokens=torch.randint(0, 4096, (B, L), generator=g),
              structure_coords=synth.synth_backbone(B, L, seed=8, nan_frac=0.3),
              chain_id=(torch.arange(L)[None] > 12).long().expand(B, L).contiguous(),
              sequence_id=(torch.arange(L)[None] // 11).expand(B, L).contiguous(),
              per_res_plddt=torch.rand(B, L, generator=g), average_plddt=torch.rand(B, L, generator=g))
    with torch.no_grad():
        r = ref(**kw)
    o = trunk_ref.esm3_forward(w, H, VH, NL, **kw)
    for k in ("embeddings", "sequence_logits", "structure_logits", "secondary_structure_logits", "sasa_logits",
              "function_logits", "residue_logits"):
        _close(o[k], getattr(r, k), 5e-5)


def test_frames_against_reference_builder():
    coords = synth.synth_backbone(3, 40, seed=77, nan_frac=0.4)
    coords[2] = float("nan")  # a row without any coordinates: identity rotation
    ref_aff, ref_mask = ref_import.build_frames(coords)
    ref_t = ref_aff.tensor  # [B, L, 12] = rotation row-major | translation (affine3d.py:386-388)
    rot, trans, mask = trunk_ref.build_frames(coords)
    assert torch.equal(mask, ref_mask)
    _close(rot.flatten(-2), ref_t[..., :9], 1e-6)
    _close(trans, ref_t[..., 9:], 1e-6)
