"""esmk_sample_track against outputs of the reference's _sample_track (tests/golden/sampling.npz: argmax and
seeded multinomial cases, with the Exp(1) noise torch.multinomial drew), and the decoding loop end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling.npz"))
FMIN = float(np.finfo(np.float32).min)


@pytest.mark.parametrize("i", range(int(G["n_sample_track"])))
def test_sample_track_matches_reference(i):
    from esm_amd.sampling import SamplingTrackConfig, sample_track
    k = f"st{i}_"
    logits = torch.from_numpy(G[k + "logits"]).cuda().contiguous()
    tokens = torch.from_numpy(G[k + "tokens"]).cuda()
    cfg = SamplingTrackConfig(temperature=float(G[k + "temperature"]), top_p=float(G[k + "top_p"]),
                              only_sample_masked_tokens=bool(G[k + "only_masked"]), invalid_ids=G[k + "invalid_ids"].tolist())
    noise = torch.from_numpy(G[k + "noise"]).cuda()
    out = sample_track(logits, tokens, cfg, int(G[k + "mask_idx"]), G[k + "valid_ids"].tolist(), noise=noise)
    assert np.array_equal(out["sampled_tokens"].cpu().numpy(), G[k + "sampled"])
    after, want_after = logits.cpu().numpy(), G[k + "logits_after"]
    assert np.array_equal(np.isneginf(after), np.isneginf(want_after))         # invalid ids
    assert np.array_equal(after == FMIN, want_after == FMIN)                     # outside the top-p nucleus
    keep = np.isfinite(want_after) & (want_after != FMIN)
    assert np.array_equal(after[keep], want_after[keep])                         # everything else untouched
    for name, tol in (("entropy", 2e-5), ("logprob", 2e-5), ("prob", 2e-6), ("top_prob", 2e-6)):
        np.testing.assert_allclose(out[name].cpu().numpy(), G[k + name], rtol=1e-4, atol=tol, err_msg=name)


def test_sampling_mask_kernel_matches_reference_semantics():
    """esmk_sampling_mask (reference sampling.py:321-342) against a plain restatement and the reference's own outputs:
    positions the reference left untouched are exactly the unmasked ones."""
    from esm_amd.sampling import SamplingTrackConfig, get_sampling_mask
    for i in range(int(G["n_sample_track"])):
        k = f"st{i}_"
        tokens = torch.from_numpy(G[k + "tokens"])
        inv, mask_idx, only = G[k + "invalid_ids"].tolist(), int(G[k + "mask_idx"]), bool(G[k + "only_masked"])
        cfg = SamplingTrackConfig(invalid_ids=inv, only_sample_masked_tokens=only)
        m = get_sampling_mask(tokens.cuda(), cfg, mask_idx).cpu()
        want = torch.ones_like(tokens, dtype=torch.bool)
        want[:, 0] = want[:, -1] = False
        for s_ in set(inv) - {mask_idx}:
            want &= tokens != s_
        if only:
            want &= tokens == mask_idx
        assert torch.equal(m, want)
        changed = torch.from_numpy(G[k + "sampled"]) != tokens
        assert not (changed & ~m).any()


def test_sasa_decoding_kernel_matches_reference():
    """esmk_sample_sasa against the outputs of the reference's sample_sasa_logits (sampling.py:260-290) and its entropy
    (generation.py:631-639) stored in tests/golden/sampling.npz."""
    from esm_amd.sampling import SamplingTrackConfig, sample_sasa_logits
    lg = torch.from_numpy(G["sasa_logits"]).clone().cuda().contiguous()
    val, ent = sample_sasa_logits(lg, torch.from_numpy(G["sasa_tokens"]).cuda(), SamplingTrackConfig(), 0, G["sasa_valid"].tolist())
    val, ent = val.cpu(), ent.cpu()
    want = torch.from_numpy(G["sasa_value"])
    assert torch.equal(torch.isinf(val), torch.isinf(want))
    fin = torch.isfinite(want)
    torch.testing.assert_close(val[fin], want[fin], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ent, torch.from_numpy(G["sasa_entropy"]), rtol=1e-5, atol=1e-5)
    # invalid classes were masked in place, like the reference's logits
    bad = [v for v in range(lg.shape[-1]) if v not in G["sasa_valid"].tolist()]
    assert torch.isneginf(lg[..., bad]).all()


def test_sample_track_errors_and_device_noise():
    from esm_amd._lib import EsmkError
    from esm_amd.sampling import SamplingTrackConfig, sample_track
    logits = torch.randn(1, 6, 64, device="cuda")
    tokens = torch.tensor([[0, 32, 32, 5, 32, 2]], device="cuda")
    with pytest.raises(ValueError):
        sample_track(logits, tokens, SamplingTrackConfig(), 32, [])
    g = torch.Generator(device="cuda").manual_seed(3)
    a = sample_track(logits.clone(), tokens, SamplingTrackConfig(temperature=1.0), 32, list(range(4, 24)), generator=g)
    g = torch.Generator(device="cuda").manual_seed(3)
    b = sample_track(logits.clone(), tokens, SamplingTrackConfig(temperature=1.0), 32, list(range(4, 24)), generator=g)
    assert torch.equal(a["sampled_tokens"], b["sampled_tokens"])
    s = a["sampled_tokens"][0]
    assert s[0] == 0 and s[3] == 5 and s[5] == 2 and all(4 <= int(s[j]) < 24 for j in (1, 2, 4))
    with pytest.raises(EsmkError):
        sample_track(torch.randn(1, 2, 20000, device="cuda"), torch.zeros(1, 2, dtype=torch.long, device="cuda"),
                     SamplingTrackConfig(temperature=0.0), 0, [1])


def test_multinomial_distribution():
    """Many rows with the same logits: empirical frequencies follow softmax(l / T)."""
    from esm_amd.sampling import SamplingTrackConfig, sample_track
    V, R = 16, 20000
    base = torch.linspace(-2, 2, V)
    logits = base.expand(1, R, V).contiguous().cuda()
    tokens = torch.full((1, R), 15, device="cuda")
    cfg = SamplingTrackConfig(temperature=0.5, only_sample_masked_tokens=False)
    g = torch.Generator(device="cuda").manual_seed(0)
    out = sample_track(logits, tokens, cfg, 15, list(range(V)), generator=g)
    ids = out["sampled_tokens"][0, 1:-1].cpu()
    freq = torch.bincount(ids, minlength=V).float() / ids.numel()
    want = torch.softmax(base / 0.5, -1)
    assert (freq - want).abs().max() < 0.01


def test_generate_sequence_with_esm3():
    from esm_amd import synth
    from esm_amd.api import ESMProteinTensor
    from esm_amd.esm3 import ESM3
    from esm_amd.sampling import GenerationConfig
    m = ESM3(128, 2, 8, 2)
    m.load_state_dict(synth.esm3_weights(128, 2, 8, seed=51))
    m = m.cuda()
    seq_a = torch.tensor([0] + [32] * 20 + [2])
    seq_b = torch.tensor([0, 5, 6, 32, 32, 32, 7, 8, 32, 2])
    struct_b = torch.tensor([4098] + [100] * 8 + [4097])
    prompts = [ESMProteinTensor(sequence=seq_a), ESMProteinTensor(sequence=seq_b, structure=struct_b)]
    cfg = GenerationConfig(track="sequence", num_steps=4, temperature=0.7)
    g = torch.Generator(device="cuda").manual_seed(1)
    out = m.generate_tokens(prompts, cfg, generator=g)
    for p, o in zip(prompts, out):
        s = o.sequence.cpu()
        assert s.shape == p.sequence.shape and not (s == 32).any()
        keep = p.sequence != 32
        assert torch.equal(s[keep], p.sequence[keep])          # prompt residues untouched
        new = s[~keep]
        assert ((new >= 4) & (new <= 30)).all()                 # only non-special vocabulary ids
    assert out[1].structure is struct_b or torch.equal(out[1].structure, struct_b)
    # entropy strategy at temperature ~0 is deterministic
    cfg2 = GenerationConfig(track="sequence", num_steps=3, temperature=0.0, temperature_annealing=False, strategy="entropy")
    o1 = m.generate_tokens(prompts, cfg2)
    o2 = m.generate_tokens(prompts, cfg2)
    assert all(torch.equal(a.sequence, b.sequence) for a, b in zip(o1, o2))
    # a missing target track starts fully masked: structure tokens for a sequence-only prompt
    cfg3 = GenerationConfig(track="structure", num_steps=2, temperature=0.0, temperature_annealing=False, strategy="entropy")
    o3 = m.generate_tokens([ESMProteinTensor(sequence=seq_b)], cfg3)[0]
    st = o3.structure.cpu()
    assert st[0] == 4098 and st[-1] == 4097 and (st[1:-1] < 4096).all()
    with pytest.raises(ValueError):
        m.generate_tokens(prompts, GenerationConfig(track="coordinates"))
    with pytest.raises(NotImplementedError):
        m.generate_tokens(prompts, GenerationConfig(track="function"))
    with pytest.raises(ValueError):
        m.generate_tokens([ESMProteinTensor(sequence=torch.tensor([0, 5, 6, 2]))], GenerationConfig(track="sequence"))


def test_generate_and_batch_generate():
    """ESM3.generate / batch_generate (reference esm3.py:377-412): per-prompt configs, error slots, num_steps
    clamped to the prompt's masks, ESMProtein prompts through encode -> tokens -> decode."""
    from esm_amd import synth
    from esm_amd.api import ESMProtein, ESMProteinError, ESMProteinTensor
    from esm_amd.esm3 import ESM3
    from esm_amd.sampling import GenerationConfig
    m = ESM3(128, 2, 8, 2)
    m.load_state_dict(synth.esm3_weights(128, 2, 8, seed=51))
    m = m.cuda()
    det = dict(temperature=0.0, temperature_annealing=False, strategy="entropy")
    seq_a = torch.tensor([0] + [32] * 20 + [2]).cuda()
    seq_b = torch.tensor([0, 5, 6, 32, 32, 32, 7, 8, 32, 2]).cuda()
    seq_c = torch.tensor([0, 5, 6, 7, 2]).cuda()  # nothing to decode
    prompts = [ESMProteinTensor(sequence=seq_a), ESMProteinTensor(sequence=seq_b), ESMProteinTensor(sequence=seq_c),
               ESMProteinTensor(sequence=seq_b.clone()), ESMProteinTensor(sequence=seq_b.clone())]
    cfgs = [GenerationConfig(track="sequence", num_steps=4, **det), GenerationConfig(track="sequence", num_steps=8, **det),
            GenerationConfig(track="sequence", num_steps=4, **det), GenerationConfig(track="structure", num_steps=2, **det),
            GenerationConfig(track="coordinates")]
    out = m.batch_generate(prompts, cfgs)
    assert len(out) == 5
    assert cfgs[1].num_steps == 4                      # 4 masked positions < 8 steps (generation.py:392-393)
    assert isinstance(out[2], ESMProteinError) and "no masks" in out[2].error_msg and out[2].error_code == 500
    assert isinstance(out[4], ESMProteinError) and "not supported" in out[4].error_msg
    for i in (0, 1):
        s_, p_ = out[i].sequence, prompts[i].sequence
        assert not (s_ == 32).any() and torch.equal(s_[p_ != 32], p_[p_ != 32])
    st = out[3].structure.cpu()
    assert st[0] == 4098 and st[-1] == 4097 and (st[1:-1] < 4096).all() and torch.equal(out[3].sequence, seq_b)
    # a prompt decoded in a batch with others == the same prompt decoded alone (deterministic settings)
    alone = m.generate(ESMProteinTensor(sequence=seq_b.clone()), GenerationConfig(track="sequence", num_steps=8, **det))
    assert (alone.sequence == out[1].sequence).float().mean() >= 0.9
    # string prompt: '_' are masks (sequence tokenizer); comes back as a string of the same length without masks
    prot = m.generate(ESMProtein(sequence="MK__LV_A"), GenerationConfig(track="sequence", num_steps=2, **det))
    assert isinstance(prot, ESMProtein) and len(prot.sequence) == 8 and "_" not in prot.sequence
    assert prot.sequence[:2] == "MK" and prot.sequence[4:6] == "LV" and prot.sequence[7] == "A"
    assert m.batch_generate([], []) == []
    with pytest.raises(AssertionError):
        m.batch_generate(prompts[:2], cfgs[:1])


def test_forward_and_sample():
    from esm_amd import synth
    from esm_amd.api import ESMProteinTensor, LogitsConfig
    from esm_amd.esm3 import ESM3
    from esm_amd.sampling import SamplingConfig, SamplingTrackConfig
    m = ESM3(128, 2, 8, 2)
    m.load_state_dict(synth.esm3_weights(128, 2, 8, seed=51))
    m = m.cuda()
    seq = torch.tensor([0, 5, 6, 32, 32, 32, 7, 8, 32, 2])
    inp = ESMProteinTensor(sequence=seq)
    cfg = SamplingConfig(sequence=SamplingTrackConfig(temperature=0.0, topk_logprobs=3),
                         structure=SamplingTrackConfig(temperature=0.0),
                         sasa=SamplingTrackConfig(), return_per_residue_embeddings=True, return_mean_embedding=True)
    out = m.forward_and_sample(inp, cfg)
    pt = out.protein_tensor
    s = pt.sequence.cpu()
    keep = seq != 32
    assert torch.equal(s[keep], seq[keep]) and not (s == 32).any()
    # argmax over the valid vocabulary of the logits the model returns for the same input
    full = ESMProteinTensor(sequence=seq.cuda(), structure=torch.tensor([4098] + [4096] * 8 + [4097]).cuda(),
                            secondary_structure=torch.zeros(10, dtype=torch.long).cuda(), sasa=torch.zeros(10, dtype=torch.long).cuda())
    lg = m.logits(full, LogitsConfig(sequence=True, structure=True)).logits
    valid = torch.zeros(64, dtype=torch.bool)
    valid[4:31] = True
    want = lg.sequence[0].cpu().masked_fill(~valid, -float("inf")).argmax(-1)
    assert torch.equal(s[~keep], want[~keep])
    st = pt.structure.cpu()
    assert st[0] == 4098 and st[-1] == 4097 and torch.equal(st[1:-1], lg.structure[0, 1:-1, :4096].cpu().argmax(-1))
    assert out.entropy.sequence.shape == (10,) and out.topk_tokens.sequence.shape == (10, 3)
    assert out.logprob.sequence[keep.cuda()].abs().max() == 0 and (out.prob.sequence <= 1.0 + 1e-6).all()
    assert pt.sasa.shape == (10,) and pt.sasa[0] == 0 and pt.sasa[-1] == 0 and out.entropy.sasa.shape == (10,)
    assert out.per_residue_embedding.shape == (10, 128)
    torch.testing.assert_close(out.mean_embedding, out.per_residue_embedding.mean(0))
    assert pt.secondary_structure.shape == (10,) and out.entropy.secondary_structure is None
    with pytest.raises(NotImplementedError):
        m.forward_and_sample(inp, SamplingConfig(function=SamplingTrackConfig()))
