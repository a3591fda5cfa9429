"""ESM3.encode / ESM3.decode over the VQ-VAE models (reference esm3.py:414-501, utils/encoding.py:48-97,
utils/decoding.py:28-169): host glue, checked against the contracts stated there; the models underneath have their own
parity tests (test_gpu_vqenc.py, test_gpu_vqvae.py)."""
import pytest
import torch

from tests import _golden as G

pytestmark = pytest.mark.gpu


def _models():
    from esm_amd import synth
    from esm_amd.esm3 import ESM3
    from esm_amd.vqvae import StructureTokenDecoder, StructureTokenEncoder
    efx, dfx, mfx = G.load("vqenc_tiny"), G.load("vqdec_tiny"), G.load("esm3_tiny_seqstruct")

    def enc_fn(device):
        m = StructureTokenEncoder(int(efx["d"]), 1, int(efx["VH"]), int(efx["NL"]), 128, int(efx["n_codes"]))
        m.load_state_dict(G.weights_for(efx), strict=True)
        return m.to(device)

    def dec_fn(device):
        m = StructureTokenDecoder(int(dfx["d"]), int(dfx["H"]), int(dfx["NL"]))
        m.load_state_dict(G.weights_for(dfx), strict=True)
        return m.to(device)

    model = ESM3(int(mfx["d"]), int(mfx["H"]), int(mfx["VH"]), int(mfx["NL"]), structure_encoder_fn=enc_fn,
                 structure_decoder_fn=dec_fn)
    model.load_state_dict(G.weights_for(mfx), strict=True)
    return model.cuda(), synth


def test_encode_forward_decode():
    from esm_amd.api import ESMProtein, ESMProteinTensor, LogitsConfig
    model, synth = _models()
    Ln = 37
    coords = synth.synth_backbone(1, Ln, seed=3, nan_frac=0.1, jitter=0.15)[0]
    coords[0], coords[-1] = coords[1] + 1.0, coords[-2] + 1.0  # synth_backbone marks BOS/EOS rows inf; a real chain has none
    seq = "ACDEFGHIKLMNPQRSTVWYACDEFGHIKLMNPQRST"
    pt = model.encode(ESMProtein(sequence=seq, coordinates=coords))
    assert pt.sequence.shape == (Ln + 2,) and int(pt.sequence[0]) == 0 and int(pt.sequence[-1]) == 2
    assert pt.structure.shape == (Ln + 2,) and int(pt.structure[0]) == 4098 and int(pt.structure[-1]) == 4097
    assert pt.coordinates.shape == (Ln + 2, 3, 3) and torch.isinf(pt.coordinates[0]).all() and torch.isinf(pt.coordinates[-1]).all()
    _, direct = model.get_structure_encoder().encode(coords[None].cuda(), residue_index=torch.arange(Ln)[None].cuda())
    assert torch.equal(pt.structure[1:-1], direct[0])
    out = model.logits(pt, LogitsConfig(sequence=True, structure=True))
    assert out.logits.sequence.shape[-2:] == (Ln + 2, 64) and torch.isfinite(out.logits.structure).all()
    prot = model.decode(ESMProteinTensor(sequence=pt.sequence, structure=pt.structure, coordinates=pt.coordinates))
    assert prot.sequence == seq
    assert prot.coordinates.shape == (Ln, 3, 3) and torch.isfinite(prot.coordinates).all()
    assert prot.plddt.shape == (Ln,) and prot.ptm.shape == (1,) and prot.pae.shape == (1, Ln + 2, Ln + 2)
    # a MASK among the structure tokens: the structure is not decoded, the given coordinates are returned (:57-61,75-76)
    st = pt.structure.clone()
    st[5] = 4096
    prot2 = model.decode(ESMProteinTensor(sequence=pt.sequence, structure=st, coordinates=pt.coordinates))
    assert prot2.plddt is None and torch.equal(prot2.coordinates.nan_to_num(), pt.coordinates[1:-1].nan_to_num())
    with pytest.raises(ValueError):
        model.encode(ESMProtein(sequence=seq[:-1], coordinates=coords))
    with pytest.raises(ValueError):
        model.encode(ESMProtein())
    # secondary-structure and SASA tracks: encode -> logits conditioned on them -> decode back
    ss = ("HHHHEEEE__CC" * Ln)[:Ln]
    sasa = [None if i % 5 == 0 else 7.5 * (i % 23) for i in range(Ln)]
    pt2 = model.encode(ESMProtein(sequence=seq, secondary_structure=ss, sasa=sasa))
    assert pt2.secondary_structure.shape == (Ln + 2,) and pt2.sasa.shape == (Ln + 2,) and pt2.sasa.is_cuda
    assert pt2.secondary_structure[0] == 0 and pt2.secondary_structure[-1] == 0 and int(pt2.secondary_structure.max()) <= 10
    lo = model.logits(pt2, LogitsConfig(sequence=True, secondary_structure=True, sasa=True))
    base = model.logits(ESMProteinTensor(sequence=pt2.sequence), LogitsConfig(sequence=True))
    assert torch.isfinite(lo.logits.sasa).all() and lo.logits.secondary_structure.shape[-1] == 11
    assert (lo.logits.sequence - base.logits.sequence).abs().max().item() > 1e-3   # the extra tracks do condition the model
    back = model.decode(pt2)
    assert back.sequence == seq
    assert back.secondary_structure == ss.replace("_", "<pad>")
    assert len(back.sasa) == Ln and all((a is None) == (b != b) for a, b in zip(sasa, back.sasa))
    # a track that is all <pad> decodes to None (decoding.py:46-56)
    empty = model.decode(ESMProteinTensor(sequence=pt2.sequence, sasa=torch.zeros(Ln + 2, dtype=torch.long)))
    assert empty.sasa is None
    assert len(model.encode(ESMProtein(secondary_structure="HHEE"))) == 6   # length inferred from the track alone


def test_missing_builders_raise():
    from esm_amd.api import ESMProtein
    from esm_amd.esm3 import ESM3
    fx = G.load("esm3_tiny_seqstruct")
    m = ESM3(int(fx["d"]), int(fx["H"]), int(fx["VH"]), int(fx["NL"])).cuda()
    with pytest.raises(ValueError):
        m.encode(ESMProtein(coordinates=torch.zeros(5, 3, 3)))
