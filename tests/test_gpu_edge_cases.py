"""Edge cases of the Python surface on the GPU: ragged / degenerate batches, the logits() wrappers and their
LogitsConfig switches, error behaviour (reference esmc.py:194-225, esm3.py:503-559, api.py:377-411)."""
import pytest
import torch

from esm_amd import synth
from esm_amd.api import ESMProteinTensor, LogitsConfig

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def esmc():
    from esm_amd.esmc import ESMC
    m = ESMC(128, 2, 3)
    m.load_state_dict(synth.esmc_weights(128, 3, seed=11))
    return m.cuda()


@pytest.fixture(scope="module")
def esm3():
    from esm_amd.esm3 import ESM3
    m = ESM3(128, 2, 8, 2)
    m.load_state_dict(synth.esm3_weights(128, 2, 8, seed=51))
    return m.cuda()


def oracle_esmc(tok, sid=None):
    from oracle import trunk_ref
    return trunk_ref.esmc_forward(tok, synth.esmc_weights(128, 3, seed=11), 2, 3, sequence_id=sid)


def close(a, b, tol=0.03):
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())


def test_single_token_and_length_one_rows(esmc):
    tok = torch.tensor([[0], [5]])
    out = esmc(tok.cuda())
    ref = oracle_esmc(tok)
    close(out.embeddings.cpu(), ref["embeddings"])
    close(out.sequence_logits.cpu(), ref["sequence_logits"])


def test_all_pad_row_and_holes_in_mask(esmc):
    tok = synth.synth_tokens([9, 5, 7], pad_to=12, seed=3)
    tok[1, :] = 1  # a row that is entirely padding
    tok[2, 3] = 1  # a hole inside a sequence: dropped from the segment, rotary position stays the column index
    out = esmc(tok.cuda())
    ref = oracle_esmc(tok)
    valid = tok != 1
    close(out.embeddings.cpu()[valid], ref["embeddings"][valid])
    assert out.embeddings.cpu()[~valid].abs().max().item() == 0.0
    assert out.hidden_states.shape == (3, 3, 12, 128)


def test_everything_padding_returns_zeros(esmc):
    tok = torch.full((2, 6), 1)
    out = esmc(tok.cuda())
    assert out.embeddings.abs().max().item() == 0.0 and out.sequence_logits.shape == (2, 6, 64)


def test_int_sequence_id_with_pads_kept(esmc):
    """An int sequence_id keeps every token (pads form their own segment and attend to each other, as in the
    reference's SDPA path, attention.py:67)."""
    tok = synth.synth_tokens([10, 6], pad_to=10, seed=4)
    sid = torch.tensor([[0] * 10, [0] * 6 + [1] * 4])
    out = esmc(tok.cuda(), sequence_id=sid.cuda())
    ref = oracle_esmc(tok, sid)
    close(out.embeddings.cpu(), ref["embeddings"])  # all positions, pads included


def test_token_range_and_noncontiguous_ids_raise(esmc):
    with pytest.raises(IndexError):
        esmc(torch.tensor([[0, 70, 2]]).cuda())
    with pytest.raises(NotImplementedError):
        esmc(torch.tensor([[0, 5, 6, 2]]).cuda(), sequence_id=torch.tensor([[0, 1, 0, 1]]).cuda())


def test_esmc_logits_config_switches(esmc):
    tok = synth.synth_tokens([17], seed=8)[0].cuda()
    pt = ESMProteinTensor(sequence=tok)
    o = esmc.logits(pt, LogitsConfig())
    assert o.logits.sequence is None and o.embeddings is None and o.hidden_states is None
    o = esmc.logits(pt, LogitsConfig(sequence=True, return_embeddings=True, return_hidden_states=True,
                                     ith_hidden_layer=1))
    assert o.logits.sequence.shape == (1, 17, 64) and o.embeddings.shape == (1, 17, 128)
    assert o.hidden_states.shape == (1, 1, 17, 128)
    full = esmc(tok[None])
    assert torch.equal(o.hidden_states[0], full.hidden_states[1])
    assert torch.equal(o.logits.sequence, full.sequence_logits)


def test_esm3_logits_matches_forward_defaults(esm3):
    """ESM3.logits: average_plddt = 1, per_res_plddt from coordinate finiteness, no sequence_id (esm3.py:515-539)."""
    from oracle import trunk_ref
    L = 21
    seq = synth.synth_tokens([L], seed=2)
    coords = synth.synth_backbone(1, L, seed=6, nan_frac=0.2)
    g = torch.Generator().manual_seed(0)
    struct = torch.randint(0, 4096, (1, L), generator=g)
    pt = ESMProteinTensor(sequence=seq[0].cuda(), structure=struct[0].cuda(), coordinates=coords[0].cuda())
    cfg = LogitsConfig(sequence=True, structure=True, secondary_structure=True, sasa=True, function=True,
                       residue_annotations=True, return_embeddings=True)
    o = esm3.logits(pt, cfg)
    w = synth.esm3_weights(128, 2, 8, seed=51)
    plddt = coords.isfinite().all(dim=-1).any(dim=-1).float()
    ref = trunk_ref.esm3_forward(w, 2, 8, 2, sequence_tokens=seq, structure_tokens=struct, structure_coords=coords,
                                 average_plddt=torch.ones(1, L), per_res_plddt=plddt)
    close(o.embeddings.cpu(), ref["embeddings"])
    close(o.logits.sequence.cpu(), ref["sequence_logits"], 0.05)
    close(o.logits.structure.cpu(), ref["structure_logits"], 0.05)
    close(o.logits.function.cpu(), ref["function_logits"], 0.05)
    close(o.residue_annotation_logits.cpu(), ref["residue_logits"], 0.05)
    assert o.logits.function.shape == (1, L, 8, 260)
    assert esm3.logits(pt, LogitsConfig()).embeddings is None


def test_esm3_requires_an_input_and_checks_ranges(esm3):
    with pytest.raises(ValueError):
        esm3()
    with pytest.raises(IndexError):
        esm3(sequence_tokens=torch.tensor([[0, 5, 2]]).cuda(), ss8_tokens=torch.tensor([[0, 11, 0]]).cuda())


def test_state_dict_reload_repacks_weights(esmc):
    tok = synth.synth_tokens([12], seed=1).cuda()
    a = esmc(tok).embeddings.clone()
    esmc.load_state_dict(synth.esmc_weights(128, 3, seed=12))
    b = esmc(tok).embeddings.clone()
    esmc.load_state_dict(synth.esmc_weights(128, 3, seed=11))
    c = esmc(tok).embeddings
    assert not torch.equal(a, b) and torch.equal(a, c)
