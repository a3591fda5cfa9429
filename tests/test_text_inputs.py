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


def test_ss8_and_sasa_tokenizers_match_reference():
    """SecondaryStructureTokenizer / SASADiscretizingTokenizer + tokenize_* / decode_* against the outputs of the
    reference's own classes (ss_tokenizer.py, sasa_tokenizer.py, encoding.py:100-135)."""
    import math

    from esm_amd.tokenization import (SASADiscretizingTokenizer, SecondaryStructureTokenizer, decode_sasa,
                                      decode_secondary_structure, tokenize_sasa, tokenize_secondary_structure)
    ss, sasa = SecondaryStructureTokenizer(), SASADiscretizingTokenizer()
    assert ss.vocab == GOLD["ss8_vocab"] and SecondaryStructureTokenizer("ss3").vocab == GOLD["ss3_vocab"]
    assert sasa.vocab == GOLD["sasa_vocab"]
    same = lambda a, b: (a is None and b is None) or (isinstance(a, float) and isinstance(b, float) and
                                                     ((math.isnan(a) and math.isnan(b)) or abs(a - b) < 1e-6))
    mids = [None if m != m else m for m in sasa.midpoints()]
    assert len(mids) == len(GOLD["sasa_midpoints"]) and all(same(a, b) for a, b in zip(mids, GOLD["sasa_midpoints"]))
    for c in GOLD["ss8"]:
        ids = tokenize_secondary_structure(c["text"], ss)
        assert ids.tolist() == c["ids"], c["text"]
        assert decode_secondary_structure(ids, ss) == c["decoded"]
    for c in GOLD["sasa"]:
        ids = tokenize_sasa(c["values"], sasa)
        assert ids.tolist() == c["ids"], c["values"]
        got = sasa.decode_float(ids[1:-1])
        assert len(got) == len(c["decode_float"]) and all(same(float(a), float(b)) for a, b in zip(got, c["decode_float"]))
        assert sasa.decode(ids[1:-1]) == c["decode"]
        assert len(decode_sasa(ids, sasa)) == len(c["values"])
    with pytest.raises(KeyError):
        ss.encode("GXH")
    with pytest.raises(ValueError):
        decode_sasa(torch.tensor([3, 4, 0]), sasa)
    assert decode_sasa(torch.tensor([0.0, 1.5, float("nan"), 0.0]), sasa) == [1.5, None]
    for t in (ss, sasa):
        assert t.mask_token_id == t.pad_token_id == t.bos_token_id == t.eos_token_id == 0 and t.special_token_ids == [0, 1, 2]
    assert len(ss.all_token_ids) == 11 and len(sasa.all_token_ids) == 19
