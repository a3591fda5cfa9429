"""Callers either side of the trunk on the GPU: text -> packs -> trunk -> pooled embeddings, against the
padded-batch ESMC.forward of the same model (itself pinned to the reference by tests/test_gpu_models.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from esm_amd import synth
    from esm_amd.esmc import ESMC
    m = ESMC(128, 2, 3)
    m.load_state_dict(synth.esmc_weights(128, 3, seed=11))
    return m.cuda()


def _seqs(n, lo=3, hi=90, seed=0):
    rng = np.random.default_rng(seed)
    return ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=int(k))) for k in rng.integers(lo, hi, size=n)]


def test_segment_mean_matches_torch():
    from esm_amd.engine import segment_mean
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 192, generator=g).cuda()
    cu = torch.tensor([0, 1, 1, 40, 171, 300], dtype=torch.int32).cuda()
    got = segment_mean(x, cu)
    for s in range(5):
        a, b = int(cu[s]), int(cu[s + 1])
        want = x[a:b].mean(0) if b > a else torch.zeros(192, device="cuda")
        torch.testing.assert_close(got[s], want, rtol=1e-5, atol=1e-6)
    # strided input (a [N, 2d] buffer's left half)
    big = torch.randn(300, 384, generator=g).cuda()
    torch.testing.assert_close(segment_mean(big[:, :192], cu)[3], big[40:171, :192].mean(0), rtol=1e-5, atol=1e-6)


def test_encode_decode_roundtrip(model):
    from esm_amd.api import ESMProtein
    pt = model.encode(ESMProtein(sequence="MKT_AC|DE"))
    assert pt.sequence.tolist() == [0, 20, 15, 11, 32, 5, 23, 31, 13, 9, 2] and pt.sequence.is_cuda
    assert model.decode(pt).sequence == "MKT_AC|DE"
    toks = model._tokenize(["ACD", "ACDEFG"])
    assert toks.shape == (2, 8) and toks[0, 5:].tolist() == [1, 1, 1]
    assert model._detokenize(toks) == ["ACD", "ACDEFG"]


def test_logits_mean_outputs(model):
    from esm_amd.api import LogitsConfig, _BatchedESMProteinTensor
    toks = model._tokenize(_seqs(5, seed=1))
    inp = _BatchedESMProteinTensor(sequence=toks)
    out = model.logits(inp, LogitsConfig(sequence=True, return_embeddings=True, return_mean_embedding=True,
                                         return_hidden_states=True, return_mean_hidden_states=True))
    valid = (toks != 1)
    for b in range(toks.shape[0]):
        want = out.embeddings[b][valid[b]].mean(0)
        torch.testing.assert_close(out.mean_embedding[b], want, rtol=1e-4, atol=1e-5)
        wh = out.hidden_states[:, b][:, valid[b]].mean(1)
        torch.testing.assert_close(out.mean_hidden_state[:, b], wh, rtol=1e-4, atol=1e-5)
    out1 = model.logits(inp, LogitsConfig(return_mean_hidden_states=True, ith_hidden_layer=1))
    assert out1.hidden_states is None and out1.mean_embedding is None
    torch.testing.assert_close(out1.mean_hidden_state, out.mean_hidden_state[1:2])


def test_embed_sequences_matches_padded_forward(model):
    from esm_amd.pipeline import embed_sequences
    seqs = _seqs(23, seed=2) + ["A", "MK_T|AC"]
    pooled = embed_sequences(model, seqs, token_budget=256)       # several packs
    toks = model._tokenize(seqs)
    ref = model.forward(toks)
    valid = toks != 1
    for i in range(len(seqs)):
        want = ref.embeddings[i][valid[i]].mean(0)
        torch.testing.assert_close(pooled[i], want, rtol=2e-2, atol=2e-2)  # bf16 trunk, different batch shapes
        cos = torch.nn.functional.cosine_similarity(pooled[i], want, dim=0)
        assert cos > 0.9999
    per_res = embed_sequences(model, seqs, token_budget=4096, pooling="none")
    for i in (0, 7, len(seqs) - 1):
        assert per_res[i].shape == (len(seqs[i]) + 2, 128)
        cos = torch.nn.functional.cosine_similarity(per_res[i], ref.embeddings[i][valid[i]], dim=-1)
        assert cos.min() > 0.999


def test_embed_fasta(model, tmp_path):
    from esm_amd.pipeline import embed_fasta, embed_sequences
    seqs = _seqs(6, seed=4)
    p = tmp_path / "in.fasta"
    p.write_text("".join(f">s{i} desc\n{s[:10]}\n{s[10:]}\n" for i, s in enumerate(seqs)))
    headers, emb = embed_fasta(model, str(p), token_budget=512)
    assert headers == [f"s{i} desc" for i in range(6)] and emb.shape == (6, 128)
    torch.testing.assert_close(emb, embed_sequences(model, seqs, token_budget=512))
    with pytest.raises(ValueError):
        embed_sequences(model, ["A" * 600], token_budget=512)
