"""CPU checks of the VQ-VAE / function-decoder host mirrors: the reference's state_dict keys load strictly (the same
dicts are loaded strictly into the REAL reference modules by oracle/make_golden.py), the bit unpacking of the function
decoder matches the oracle, and there is no CPU fallback."""
import pytest
import torch

from esm_amd import synth
from esm_amd._lib import EsmkError
from esm_amd.function_decoder import FunctionTokenDecoder, FunctionTokenDecoderConfig
from esm_amd.vqvae import StructureTokenDecoder, StructureTokenEncoder
from oracle import function_decoder_ref


def test_state_dict_keys_and_no_cpu_path():
    dec = StructureTokenDecoder(128, 2, 2)
    w = synth.make_weights(synth.vqdec_shapes(128, 2), seed=1)
    assert set(dec.state_dict()) == set(w)
    dec.load_state_dict(w, strict=True)
    tok = torch.tensor([[4098, 5, 7, 4097]])
    with pytest.raises(EsmkError):
        dec.decode(tok)
    enc = StructureTokenEncoder(128, 1, 8, 2, 128, 512)
    w = synth.vqenc_weights(128, 8, 2, 128, 512, seed=1)
    assert set(enc.state_dict()) == set(w)
    enc.load_state_dict(w, strict=True)
    with pytest.raises(EsmkError):
        enc.encode(torch.zeros(1, 20, 3, 3))
    cfg = FunctionTokenDecoderConfig(d_model=256, n_heads=2, n_layers=2, num_interpro_classes=200, keyword_vocabulary_size=300)
    fd = FunctionTokenDecoder(cfg)
    w = synth.fdec_weights(256, 2, 300, 200, seed=1)
    assert set(fd.state_dict()) == set(w)
    fd.load_state_dict(w, strict=True)
    with pytest.raises(EsmkError):
        fd(torch.full((3, 8), 10))


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        FunctionTokenDecoder(FunctionTokenDecoderConfig(unpack_lsh_bits=False))
    with pytest.raises(NotImplementedError):
        FunctionTokenDecoder(FunctionTokenDecoderConfig(d_model=1024, n_heads=16))


def test_function_token_bit_unpacking_matches_oracle():
    fd = FunctionTokenDecoder(FunctionTokenDecoderConfig(d_model=256, n_heads=2, n_layers=1, num_interpro_classes=8,
                                                         keyword_vocabulary_size=8))
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, 260, (50, 8), generator=g)
    got = fd._inputs(tok)
    assert torch.equal(got, function_decoder_ref.unpack_inputs(tok))
    assert got.shape == (50, 64) and int(got.max()) < 128
    special = (tok < 4).any(1)
    assert special.any() and bool((got[special] == 0).all())


def test_decoder_input_assertions_precede_the_device_check():
    dec = StructureTokenDecoder(128, 2, 1)
    with pytest.raises(AssertionError):
        dec.decode(torch.tensor([[5, 5, 7, 4097]]))       # no BOS
    with pytest.raises(AssertionError):
        dec.decode(torch.tensor([[4098, 5, 7, 9]]))       # no EOS
    with pytest.raises(AssertionError):
        dec.decode(torch.tensor([[4098, -1, 7, 4097]]))   # unresolved -1
