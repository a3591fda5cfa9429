"""Input side (CPU): tokenizer and FASTA parser against fixtures produced by the reference's own
EsmSequenceTokenizer / tokenize_sequence / parse_fasta (oracle/make_golden.py text_cases), plus the pack planner."""
import io
import json
import os

import numpy as np
import pytest
import torch

from esm_amd.parsing import parse_fasta, read_first_sequence, read_sequences
from esm_amd.pipeline import gather_pack, tokenize_packs
from esm_amd.tokenization import (EsmSequenceTokenizer, decode_sequence, stack_variable_length_tensors,
                                  tokenize_sequence)

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_inputs.json")))


def test_tokenizer_ids_match_reference():
    tok = EsmSequenceTokenizer()
    for k, v in GOLD["tokenizer_attrs"].items():
        got = getattr(tok, k)
        assert (sorted(got) if k == "special_token_ids" else got) == v, k
    for c in GOLD["tokenizer"]:
        assert tokenize_sequence(c["text"], tok).tolist() == c["ids"], c["text"]
        assert tokenize_sequence(c["text"], tok, add_special_tokens=False).tolist() == c["ids_nospecial"], c["text"]
        assert tok.decode(c["ids"]) == c["hf_decode"]


def test_encode_packed_equals_per_sequence():
    tok = EsmSequenceTokenizer()
    texts = [c["text"] for c in GOLD["tokenizer"]]
    flat, lens = tok.encode_packed(texts)
    assert flat.tolist() == sum((c["ids"] for c in GOLD["tokenizer"]), [])
    assert lens.tolist() == [len(c["ids"]) for c in GOLD["tokenizer"]]
    plain = [t for t in texts if "<" not in t and t.isascii()]  # the vectorised path alone
    flat2, lens2 = tok.encode_packed(plain)
    want = [c["ids"] for c in GOLD["tokenizer"] if c["text"] in plain]
    assert flat2.tolist() == sum(want, []) and lens2.tolist() == [len(w) for w in want]
    body, blen = tok.encode_packed(plain, add_special_tokens=False)
    assert body.tolist() == sum((w[1:-1] for w in want), []) and blen.tolist() == [len(w) - 2 for w in want]
    e, el = tok.encode_packed([])
    assert e.shape == (0,) and el.shape == (0,)


def test_decode_sequence_roundtrip():
    tok = EsmSequenceTokenizer()
    for s in ("ACDEFGHIKLMNPQRSTVWY", "AC_D|EF", "BUZO.-X"):
        assert decode_sequence(tokenize_sequence(s, tok), tok) == s
    assert decode_sequence(torch.tensor([0, 5, 3, 1, 2]), tok) == "A<unk><pad>"


def test_stack_variable_length():
    a, b = torch.tensor([0, 5, 2]), torch.tensor([0, 5, 6, 7, 2])
    out = stack_variable_length_tensors([a, b], constant_value=1)
    assert out.tolist() == [[0, 5, 2, 1, 1], [0, 5, 6, 7, 2]]


def test_fasta_matches_reference(tmp_path):
    got = [[e.header, e.sequence] for e in parse_fasta(GOLD["fasta_text"])]
    assert got == GOLD["fasta_entries"]
    p = tmp_path / "x.fasta"
    p.write_text(GOLD["fasta_text"])
    assert [[e.header, e.sequence] for e in read_sequences(str(p))] == GOLD["fasta_entries"]
    assert [[e.header, e.sequence] for e in read_sequences(io.StringIO(GOLD["fasta_text"]))] == GOLD["fasta_entries"]
    import gzip
    with gzip.open(tmp_path / "x.fasta.gz", "wt") as f:
        f.write(GOLD["fasta_text"])
    assert read_first_sequence(tmp_path / "x.fasta.gz").header == GOLD["fasta_entries"][0][0]
    with pytest.raises(ValueError):
        list(parse_fasta("# nothing here\n\n"))


def test_tokenize_packs_covers_every_sequence_once():
    rng = np.random.default_rng(3)
    seqs = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), size=int(n))) for n in rng.integers(5, 300, size=57)]
    tok = EsmSequenceTokenizer()
    tokens, lengths, offsets, packs, per_rank = tokenize_packs(seqs, tok, world=3, token_budget=1024, d=128, h=512)
    assert lengths.tolist() == [len(s) + 2 for s in seqs] and offsets[-1] == tokens.shape[0]
    seen = sorted(i for r in per_rank for p in r for i in packs[p])
    assert seen == list(range(len(seqs)))
    assert sorted(p for r in per_rank for p in r) == list(range(len(packs)))
    buf = np.empty(1024, dtype=np.int64)
    for p in packs:
        n = gather_pack(tokens, offsets, p, buf)
        assert n == sum(lengths[i] for i in p) <= 1024
        assert buf[:n].tolist() == sum((tokenize_sequence(seqs[i], tok).tolist() for i in p), [])


def test_save_load_embeddings(tmp_path):
    from esm_amd.pipeline import load_embeddings, save_embeddings
    emb = torch.randn(3, 16)
    save_embeddings(tmp_path / "e.safetensors", ["a b", "c", "d|e"], emb)
    h, e = load_embeddings(tmp_path / "e.safetensors")
    assert h == ["a b", "c", "d|e"] and torch.equal(e, emb)
    save_embeddings(tmp_path / "h.safetensors", ["x"], emb[:1], dtype=torch.bfloat16)
    assert load_embeddings(tmp_path / "h.safetensors")[1].dtype == torch.bfloat16
    with pytest.raises(ValueError):
        save_embeddings(tmp_path / "bad.safetensors", ["only one"], emb)
