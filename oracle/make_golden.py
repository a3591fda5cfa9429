"""Generate tests/golden/*.npz by running the IMPORTED reference (fp32, CPU) on seeded inputs.

Run only in the build container:  python -m oracle.make_golden
Weights are not stored: every fixture records (model kind, dims, weight seed) and the tests
regenerate identical weights with esm_amd.synth.  Stored: inputs + reference outputs (fp32,
sub-sampled where large).  A fixture is data only - no reference source travels.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from esm_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def esmc_case(name, d, H, NL, lengths, wseed, tseed, sequence_id=None, pad_to=None, store_hidden=(0, -1),
              token_stride=1):
    w = synth.esmc_weights(d, NL, seed=wseed)
    m = ref_import.build_esmc(d, H, NL, w)
    tok = synth.synth_tokens(lengths, pad_to=pad_to, seed=tseed)
    with torch.no_grad():
        out = m(tok, sequence_id=sequence_id)
    arrays = dict(
        kind="esmc", d=d, H=H, NL=NL, wseed=wseed, tokens=_np(tok), token_stride=token_stride,
        sequence_logits=_np(out.sequence_logits[:, ::token_stride]),
        embeddings=_np(out.embeddings[:, ::token_stride]),
        hidden_layers=np.asarray([i % NL for i in store_hidden]),
        hidden_states=_np(out.hidden_states[[i % NL for i in store_hidden]][:, :, ::token_stride]),
    )
    if sequence_id is not None:
        arrays["sequence_id"] = _np(sequence_id)
    if NL >= 30:  # calibration: the reference's OWN bf16 path against its fp32 path on these inputs
        mb = m.to(torch.bfloat16)
        with torch.no_grad():
            ob = mb(tok, sequence_id=sequence_id)
        valid = tok != 1
        cos = torch.nn.functional.cosine_similarity(ob.embeddings.float()[valid], out.embeddings[valid], dim=-1)
        arrays["ref_bf16_cos_min"] = float(cos.min())
        arrays["ref_bf16_logit_maxdiff"] = float((ob.sequence_logits.float()[valid] - out.sequence_logits[valid]).abs().max())
        arrays["ref_logit_absmax"] = float(out.sequence_logits[valid].abs().max())
        print(name, "reference bf16 vs fp32: cos_min", arrays["ref_bf16_cos_min"], "logit maxdiff", arrays["ref_bf16_logit_maxdiff"],
              "of", arrays["ref_logit_absmax"])
    save(name, **arrays)


def packed_row(lengths, total):
    """One bin-packed row: tokens of several sequences back-to-back + int64 sequence_id."""
    toks = [synth.synth_tokens([n], seed=900 + i)[0] for i, n in enumerate(lengths)]
    row = torch.full((total,), 1, dtype=torch.int64)
    sid = torch.full((total,), len(lengths), dtype=torch.int64)  # trailing pads form their own id
    pos = 0
    for i, t in enumerate(toks):
        row[pos: pos + len(t)] = t
        sid[pos: pos + len(t)] = i
        pos += len(t)
    return row, sid


def stack_case(name, d, H, NL, L, wseed, scale_residue=True):
    """True-width trunk, few layers: input activations -> (post, pre)."""
    w = synth.make_weights(synth.trunk_shapes("", d, NL), seed=wseed)
    m = ref_import.build_stack(d, H, NL, w, scale_residue=scale_residue)
    g = torch.Generator().manual_seed(wseed + 17)
    x = torch.randn(2, L, d, generator=g)
    sid = torch.ones(2, L, dtype=torch.bool)
    sid[1, L - 9:] = False
    with torch.no_grad():
        post, pre, hid = m(x, sequence_id=sid)
    save(name, kind="stack", d=d, H=H, NL=NL, wseed=wseed, x=_np(x), sequence_id=_np(sid),
         post=_np(post[:, ::4]), pre=_np(pre[:, ::4]), token_stride=4, scale_residue=int(scale_residue))


def stack_long_case(name, d, H, NL, L, wseed, row_stride, n_pad=100):
    """True-width trunk at a long sequence (config C5: d2560 / 40 heads at L = 4096): one row, the last n_pad positions
    masked out.  The input is regenerated from its seed by the test (synth.synth_activations), outputs are stored
    for every row_stride-th position."""
    w = synth.make_weights(synth.trunk_shapes("", d, NL), seed=wseed)
    m = ref_import.build_stack(d, H, NL, w)
    x = synth.synth_activations((1, L, d), seed=wseed + 17)
    sid = torch.ones(1, L, dtype=torch.bool)
    sid[0, L - n_pad:] = False
    with torch.no_grad():
        post, pre, hid = m(x, sequence_id=sid)
    save(name, kind="stack", d=d, H=H, NL=NL, wseed=wseed, x_seed=wseed + 17, x_shape=np.asarray([1, L, d]),
         sequence_id=_np(sid), post=_np(post[:, ::row_stride]), pre=_np(pre[:, ::row_stride]), token_stride=row_stride,
         scale_residue=1)


def fullsize_cases():
    """BASELINE.json configs C3 and C5 at their full sequence lengths (round 2): the 256-head geometric attention on
    512-key rows, the ESM3-open-shaped model at full depth on one 512-residue protein, and the d2560 trunk at L = 4096."""
    geom_case("geom_d1536_vh256_L512", 1536, 256, 1, 512, wseed=44, variant="multichain", nan_frac=0.1, row_stride=8)
    stack_long_case("stack_d2560_L4096", 2560, 40, 2, 4096, wseed=71, row_stride=64)
    esm3_case("esm3_open_48l_L512", 1536, 24, 256, 48, 1, 512, [512], wseed=54, iseed=504, with_tracks=False,
              big_stride=32, emb_stride=8)


def stack_variant_case(name, d, H, NL, L, wseed, VH=None, **kw):
    """TransformerStack with the other block variants of blocks.py:73-115 (bias / no QK-LayerNorm / GELU FFN /
    geometric attention without frameless zeroing): input activations (+ coordinates, chains) -> (post, pre).
    Weights: synth.stack_variant_weights over the reference module's own state_dict shapes."""
    m = ref_import.build_stack_variant(d, H, NL, VH, **kw)
    w = synth.stack_variant_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=wseed)
    m.load_state_dict(w, strict=True)
    g = torch.Generator().manual_seed(wseed + 17)
    x = torch.randn(2, L, d, generator=g)
    sid = torch.zeros(2, L, dtype=torch.int64)
    sid[0, L // 2:] = 1          # row 0: two bin-packed sequences
    sid[1, L - 7:] = 5           # row 1: one sequence + a run of padding with its own id
    arrays = dict(kind="stack_variant", d=d, H=H, NL=NL, VH=VH or 0, wseed=wseed, x=_np(x), sequence_id=_np(sid),
                  ctor=repr(sorted(kw.items())))
    affine = mask = chain = None
    if kw.get("n_layers_geom", 1) > 0:
        from esm.utils.structure.affine3d import build_affine3d_from_coordinates
        coords = synth.synth_backbone(2, L, seed=wseed + 4, nan_frac=0.2)
        chain = torch.zeros(2, L, dtype=torch.int64)
        chain[1, L // 3:] = 1
        affine, mask = build_affine3d_from_coordinates(coords)
        arrays.update(coords=_np(coords), chain_id=_np(chain))
    with torch.no_grad():
        post, pre, hid = m(x, sequence_id=sid, affine=affine, affine_mask=mask, chain_id=chain)
    save(name, post=_np(post), pre=_np(pre), hidden0=_np(hid[0]), **arrays)


STACK_VARIANTS = {  # name -> (d, H, NL, L, wseed, VH, constructor options)
    "stackvar_bias_noqkln_gelu": (128, 2, 2, 40, 81, None, dict(n_layers_geom=0, bias=True, qk_layernorm=False,
                                                                  ffn_type="gelu", expansion_ratio=4.0)),
    "stackvar_noqkln_swiglu": (128, 2, 3, 40, 82, None, dict(n_layers_geom=0, qk_layernorm=False)),
    "stackvar_gelu_noscale": (192, 3, 2, 33, 83, None, dict(n_layers_geom=0, ffn_type="gelu", expansion_ratio=2.0,
                                                            scale_residue=False)),
    "stackvar_geom_nozero": (128, 2, 2, 40, 84, 8, dict(n_layers_geom=1, mask_and_zero_frameless=False)),
    "stackvar_bias_qkln": (128, 2, 2, 40, 86, None, dict(n_layers_geom=0, bias=True, qk_layernorm=True)),
    "stackvar_bias_geom2": (128, 2, 2, 40, 85, 8, dict(n_layers_geom=2, bias=True, qk_layernorm=False, ffn_type="gelu",
                                                        expansion_ratio=4.0, mask_and_zero_frameless=True)),
}


def stack_variant_cases():
    for name, (d, H, NL, L, wseed, VH, kw) in STACK_VARIANTS.items():
        stack_variant_case(name, d, H, NL, L, wseed, VH=VH, **kw)


def vqdec_tokens(lengths, L, seed):
    """Structure tokens [B, L]: BOS, codes (with one MASK and one CHAINBREAK inside row 0), EOS, PAD after."""
    g = torch.Generator().manual_seed(seed)
    tok = torch.full((len(lengths), L), 4099, dtype=torch.int64)
    for b, n in enumerate(lengths):
        tok[b, :n] = torch.randint(0, 4096, (n,), generator=g)
        tok[b, 0], tok[b, n - 1] = 4098, 4097
    tok[0, 5], tok[0, 9] = 4096, 4100
    return tok, tok != 4099


def vqdec_case(name, d, H, NL, L, lengths, wseed, tseed, sequence_id=None, pae_stride=1):
    """StructureTokenDecoder.decode end to end (vqvae.py:365-437), plus the reference's own bf16 drift."""
    w = synth.make_weights(synth.vqdec_shapes(d, NL), seed=wseed)
    m = ref_import.build_vqdec(d, H, NL, w)
    tok, mask = vqdec_tokens(lengths, L, tseed)
    with torch.no_grad():
        out = m.decode(tok, attention_mask=mask, sequence_id=sequence_id)
    arrays = dict(kind="vqdec", d=d, H=H, NL=NL, wseed=wseed, tokens=_np(tok), attention_mask=_np(mask))
    if sequence_id is not None:
        arrays["sequence_id"] = _np(sequence_id)
    for k, v in out.items():
        arrays[k] = _np(v[:, ::pae_stride, ::pae_stride]) if k == "predicted_aligned_error" else _np(v)
    arrays["pae_stride"] = pae_stride
    mb = m.to(torch.bfloat16)
    with torch.no_grad():
        ob = mb.decode(tok, attention_mask=mask, sequence_id=sequence_id)
    for k, v in out.items():
        arrays["ref_bf16_maxdiff_" + k] = float((ob[k].float() - v).abs().max())
        arrays["ref_bf16_meandiff_" + k] = float((ob[k].float() - v).abs().mean())
        print(name, k, "absmax", float(v.abs().max()), "reference bf16 drift max", arrays["ref_bf16_maxdiff_" + k],
              "mean", arrays["ref_bf16_meandiff_" + k])
    save(name, **arrays)


def esm3_inputs(B, L, lengths, seed, with_tracks=True):
    g = torch.Generator().manual_seed(seed)
    seq = synth.synth_tokens(lengths, pad_to=L, seed=seed)
    kw = dict(sequence_tokens=seq)
    kw["structure_tokens"] = torch.randint(0, 4096, (B, L), generator=g)
    coords = synth.synth_backbone(B, L, seed=seed + 1, nan_frac=0.15)
    for b, n in enumerate(lengths):
        coords[b, n - 1] = float("inf")  # EOS row
        coords[b, n:] = float("nan")  # pad rows
    kw["structure_coords"] = coords
    if with_tracks:
        kw["ss8_tokens"] = torch.randint(0, 11, (B, L), generator=g)
        kw["sasa_tokens"] = torch.randint(0, 19, (B, L), generator=g)
        f = torch.randint(0, 260, (B, L, 8), generator=g)
        f[:, : L // 3] = 0
        kw["function_tokens"] = f
        r = torch.randint(0, 1478, (B, L, 16), generator=g)
        r[:, :, 3:] = 0
        r[:, ::2] = 0
        kw["residue_annotation_tokens"] = r
        kw["average_plddt"] = torch.rand(B, L, generator=g)
        kw["per_res_plddt"] = torch.rand(B, L, generator=g)
    return kw


def esm3_case(name, d, H, VH, NL, B, L, lengths, wseed, iseed, extra=None, with_tracks=True, big_stride=1,
              emb_stride=1):
    w = synth.esm3_weights(d, NL, VH, seed=wseed)
    m = ref_import.build_esm3(d, H, VH, NL, w)
    kw = esm3_inputs(B, L, lengths, iseed, with_tracks)
    if extra:
        kw.update(extra)
    with torch.no_grad():
        out = m(**kw)
    arrays = dict(kind="esm3", d=d, H=H, VH=VH, NL=NL, wseed=wseed, big_stride=big_stride, emb_stride=emb_stride)
    for k, v in kw.items():
        arrays["in_" + k] = _np(v)
    for k in ("sequence_logits", "secondary_structure_logits", "sasa_logits"):
        arrays[k] = _np(getattr(out, k))
    arrays["embeddings"] = _np(out.embeddings[:, ::emb_stride])
    for k in ("structure_logits", "function_logits", "residue_logits"):
        arrays[k] = _np(getattr(out, k)[:, ::big_stride])
    save(name, **arrays)


def geom_case(name, d, VH, B, L, wseed, variant, nan_frac=0.2, row_stride=1):
    """GeometricReasoningOriginalImpl alone (block 0 of a 1-layer geom stack shares the weights).  row_stride > 1:
    the input s is NOT stored (the test regenerates it with synth.synth_activations(seed = wseed + 3)) and only every
    row_stride-th output row is."""
    ref_import.install()
    from esm.layers.geom_attention import GeometricReasoningOriginalImpl
    from esm.utils.structure.affine3d import build_affine3d_from_coordinates

    shapes = {k[len("blocks.0.geom_attn."):]: v for k, v in
              synth.trunk_shapes("", d, 1, v_heads=VH, n_layers_geom=1).items() if "geom_attn" in k}
    w = synth.make_weights({"blocks.0.geom_attn." + k: v for k, v in shapes.items()}, seed=wseed)
    mod = GeometricReasoningOriginalImpl(c_s=d, v_heads=VH, mask_and_zero_frameless=True).eval()
    mod.load_state_dict({k[len("blocks.0.geom_attn."):]: v for k, v in w.items()}, strict=True)
    if row_stride > 1:
        s = synth.synth_activations((B, L, d), seed=wseed + 3)
    else:
        g = torch.Generator().manual_seed(wseed + 3)
        s = torch.randn(B, L, d, generator=g)
    coords = synth.synth_backbone(B, L, seed=wseed + 4, nan_frac=nan_frac)
    sid = None
    chain = torch.zeros(B, L, dtype=torch.int64)
    if variant == "nostruct":
        coords[:] = float("nan")
    elif variant == "multichain":
        chain[:, L // 2:] = 1
    elif variant == "binpacked":
        sid = torch.zeros(B, L, dtype=torch.int64)
        sid[:, L // 3:] = 1
        sid[:, 2 * L // 3:] = 2
    elif variant == "onerow_frameless":
        coords[0] = float("nan")
    affine, mask = build_affine3d_from_coordinates(coords)
    with torch.no_grad():
        out = mod(s, affine, mask, sid, chain)
    arrays = dict(kind="geom", d=d, VH=VH, wseed=wseed, coords=_np(coords), chain_id=_np(chain),
                  out=_np(out[:, ::row_stride]), affine=_np(affine.tensor), affine_mask=_np(mask), row_stride=row_stride)
    if row_stride > 1:
        arrays.update(s_seed=wseed + 3, s_shape=np.asarray([B, L, d]))
    else:
        arrays["s"] = _np(s)
    if sid is not None:
        arrays["sequence_id"] = _np(sid)
    save(name, **arrays)


def frames_case(name):
    B, L = 3, 40
    coords = synth.synth_backbone(B, L, seed=77, nan_frac=0.25)
    coords[1] = float("nan")  # a batch row with no coordinates at all -> identity rotation
    coords[2, 5, 1, 0] = 2e6  # beyond _MAX_SUPPORTED_DISTANCE -> frameless
    affine, mask = ref_import.build_frames(coords)
    save(name, kind="frames", coords=_np(coords), affine=_np(affine.tensor), mask=_np(mask))


TOKENIZER_STRINGS = [
    "ACDEFGHIKLMNPQRSTVWY", "ACD_EF", "AC|DE", "acdX", "A C", "BUZO.-", "<mask>AC<eos>", "", "J", "A<unk>C",
    "\u00e9A", "AC<pad>D", "<cls>A", "A\nC", "A<maskC", "__", "MKT|_LLV<mask>", "X" * 70,
]
FASTA_TEXT = """# comment line
>sp|P1 first protein
MKTAYIAK
QRQISFVK

>second
ACD_EF|GH
>empty_record
>last  \t
  WW
"""


def text_cases():
    """Input-side fixtures: the reference tokenizer (sequence_tokenizer.py + encoding.py:35-45 + decoding.py:125-135
    restated: utils/decoding cannot be imported here, circular import through esm.sdk) and FASTA parser."""
    import json
    ref_import.install()
    from esm.tokenization.sequence_tokenizer import EsmSequenceTokenizer
    from esm.utils import encoding
    from esm.utils.parsing import parse_fasta
    tok = EsmSequenceTokenizer()
    cases = []
    for s in TOKENIZER_STRINGS:
        ids = encoding.tokenize_sequence(s, tok).tolist()
        cases.append({"text": s, "ids": ids, "ids_nospecial": encoding.tokenize_sequence(s, tok, add_special_tokens=False).tolist(),
                      "hf_decode": tok.decode(ids)})
    fasta = [[e.header, e.sequence] for e in parse_fasta(FASTA_TEXT)]
    attrs = {k: getattr(tok, k) for k in ("pad_token_id", "mask_token_id", "cls_token_id", "eos_token_id", "bos_token_id",
                                          "unk_token_id", "vocab_size")}
    attrs["chain_break_token_id"] = tok.convert_tokens_to_ids("|")
    attrs["all_token_ids"] = list(tok.all_token_ids)
    attrs["special_token_ids"] = sorted(tok.special_token_ids)
    # secondary-structure and SASA tokenizers (ss_tokenizer.py, sasa_tokenizer.py, encoding.py:100-135)
    from esm.tokenization.sasa_tokenizer import SASADiscretizingTokenizer
    from esm.tokenization.ss_tokenizer import SecondaryStructureTokenizer
    ss_tok, sasa_tok = SecondaryStructureTokenizer(), SASADiscretizingTokenizer()
    ss_cases = []
    for text in ("GHITEBSC", "CC_HH__EE", "", "<pad>HH<pad>E", "_"):
        ids = encoding.tokenize_secondary_structure(text, ss_tok)
        ss_cases.append({"text": text, "ids": ids.tolist(), "decoded": ss_tok.decode(ids[1:-1])})
    sasa_cases = []
    for vals in ([0.0, 0.8, 0.81, 3.99, 4.0, 150.0, 151.4, 151.5, 1e6], [None, 12.5, "<unk>", 33, None], [], [None]):
        ids = encoding.tokenize_sasa(vals, sasa_tok)
        sasa_cases.append({"values": vals, "ids": ids.tolist(), "decode_float": sasa_tok.decode_float(ids[1:-1]),
                           "decode": sasa_tok.decode(ids[1:-1])})
    extra = {"ss8": ss_cases, "ss8_vocab": ss_tok.vocab, "ss3_vocab": SecondaryStructureTokenizer("ss3").vocab,
             "sasa": sasa_cases, "sasa_vocab": sasa_tok.vocab, "sasa_midpoints": [None if m != m else m for m in sasa_tok.midpoints()]}
    with open(os.path.join(OUT, "text_inputs.json"), "w") as f:
        json.dump({"tokenizer": cases, "tokenizer_attrs": attrs, "fasta_text": FASTA_TEXT, "fasta_entries": fasta, **extra}, f, indent=1)
    print("text_inputs.json written")


def sampling_cases():
    """Sampler fixtures from the reference's own _sample_track / step-mask / annealing functions
    (esm/utils/generation.py:241-352,703-822, esm/utils/sampling.py:153-343)."""
    ref_import.install()
    import esm.sdk.api as api  # noqa: F401  (must precede esm.utils.generation: circular import otherwise)
    from esm.sdk.api import GenerationConfig, SamplingTrackConfig
    from esm.utils import generation as G
    from esm.utils.sampling import _BatchedESMProteinTensor
    from esm.tokenization.sequence_tokenizer import EsmSequenceTokenizer
    from esm.tokenization.structure_tokenizer import StructureTokenizer
    from types import SimpleNamespace
    out = {}
    g = torch.Generator().manual_seed(7)
    n_case = 0
    for V, mask_idx, specials in ((64, 32, [0, 1, 2, 31, 32]), (4101, 4096, [4096, 4097, 4098, 4099, 4100])):
        for temperature, top_p, only_masked, seed in ((0.0, 1.0, True, 0), (0.0, 0.6, False, 0), (0.7, 1.0, True, 11),
                                                      (1.3, 0.8, True, 12)):
            B, L = (2, 24) if V == 64 else (2, 10)  # keep the 4101-wide fixtures small
            logits = torch.randn(B, L, V, generator=g) * 3.0
            tokens = torch.randint(4, 24 if V == 64 else 4000, (B, L), generator=g)
            tokens[:, 0] = specials[0] if V == 64 else 4098
            tokens[:, -1] = specials[2] if V == 64 else 4097
            tokens[0, 3:L - 2] = mask_idx
            tokens[1, 5:9] = mask_idx
            tokens[1, L - 3] = specials[1]  # a pad / special inside the row
            invalid = [5, 9] if V == 64 else [17, 300]
            valid_ids = sorted(set(range(V if V == 64 else 4101)) - set(specials) - set(invalid))
            if V == 64:
                valid_ids = [v for v in valid_ids if v < 33]  # tokenizer vocabulary is 33 wide, logits 64
            cfg = SamplingTrackConfig(temperature=temperature, top_p=top_p, only_sample_masked_tokens=only_masked,
                                      invalid_ids=invalid, topk_logprobs=0)
            lg = logits.clone()
            torch.manual_seed(seed)
            meta = G._sample_track(lg, tokens, cfg, mask_idx, valid_ids)
            torch.manual_seed(seed)
            noise = torch.empty(B * L, V).exponential_(1)  # what torch.multinomial draws first (n_sample == 1 path)
            if temperature > 0:  # pin the noise recipe: multinomial(probs, 1) == argmax(probs / q)
                probs = torch.softmax(lg.reshape(-1, V) / temperature, -1)
                smask = G.get_sampling_mask(tokens, cfg, mask_idx).reshape(-1)
                redraw = (probs / noise).argmax(-1)
                assert torch.equal(redraw[smask], meta["sampled_tokens"].reshape(-1)[smask]), "noise recipe mismatch"
            k = f"st{n_case}_"
            out.update({k + "logits": _np(logits), k + "tokens": _np(tokens), k + "valid_ids": np.asarray(valid_ids),
                        k + "invalid_ids": np.asarray(invalid), k + "mask_idx": mask_idx, k + "temperature": temperature,
                        k + "top_p": top_p, k + "only_masked": only_masked, k + "noise": _np(noise),
                        k + "logits_after": _np(lg), k + "sampled": _np(meta["sampled_tokens"]),
                        k + "entropy": _np(meta["entropy"]), k + "logprob": _np(meta["logprob"]),
                        k + "prob": _np(meta["prob"]), k + "top_prob": _np(meta["top_prob"])})
            n_case += 1
    out["n_sample_track"] = n_case
    # SASA decoding to floats (sampling.py:255-290) and its entropy (generation.py:631-639)
    from esm.utils.sampling import sample_sasa_logits
    lg = torch.randn(2, 14, 19, generator=g) * 2.0
    lg[0, 4, 18] = 30.0  # argmax on the last class -> inf
    tk = torch.zeros(2, 14, dtype=torch.int64)
    tk[1, 3:6] = torch.tensor([5, 7, 9])
    scfg = SamplingTrackConfig(only_sample_masked_tokens=True, invalid_ids=[])
    valid = list(range(3, 19))
    lg_in = lg.clone()
    val = sample_sasa_logits(lg, tk, scfg, 0, valid)
    ent = -(torch.nan_to_num(lg.softmax(-1) * lg.log_softmax(-1))).sum(-1)
    out.update({"sasa_logits": _np(lg_in), "sasa_tokens": _np(tk), "sasa_valid": np.asarray(valid), "sasa_value": _np(val),
                "sasa_entropy": _np(ent)})
    # step masks: which masked positions are unmasked at step t (entropy strategy is deterministic)
    toks = SimpleNamespace(sequence=EsmSequenceTokenizer(), structure=StructureTokenizer())
    n_mask = 0
    for track, mask_idx, L, n_masked, num_steps, schedule in (("sequence", 32, 40, 30, 8, "cosine"),
                                                              ("structure", 4096, 33, 31, 5, "linear"),
                                                              ("sequence", 32, 20, 3, 3, "cosine")):
        tokens = torch.randint(4, 24, (1, L), generator=g)
        tokens[0, 0], tokens[0, -1] = (0, 2) if track == "sequence" else (4098, 4097)
        pos = torch.randperm(L - 2, generator=g)[:n_masked] + 1
        tokens[0, pos] = mask_idx
        cur = tokens.clone()
        ent = torch.rand(1, L, generator=g)
        cfg = GenerationConfig(track=track, num_steps=num_steps, schedule=schedule, strategy="entropy")
        wheres = []
        for t in range(num_steps):
            prot = _BatchedESMProteinTensor(**{track: cur})
            w = G._get_iterative_sampling_mask_for_prompt_and_step(
                prot, torch.tensor(L), torch.tensor(n_masked), t, api.ForwardTrackData(**{track: ent}), cfg, toks)
            wheres.append(_np(w))
            cur = torch.where(w, torch.full_like(cur, 5), cur)
        k = f"sm{n_mask}_"
        out.update({k + "track": track, k + "mask_idx": mask_idx, k + "tokens": _np(tokens), k + "entropy": _np(ent),
                    k + "num_steps": num_steps, k + "schedule": schedule, k + "n_masked": n_masked,
                    k + "where": np.stack(wheres)})
        n_mask += 1
    out["n_step_mask"] = n_mask
    out["annealed"] = np.asarray([[G._get_annealed_temperature(t, n, T) for t in range(n)]
                                  for n, T in ((1, 1.0), (8, 0.7), (8, 1.5))][1:], dtype=np.float64)
    out["annealed_n1"] = G._get_annealed_temperature(0, 1, 1.0)
    save("sampling", **out)


def vqenc_case(name, d, VH, NL, n_codes, B, L, wseed, cseed, nan_frac=0.1, pad_from=None, sequence_id=None,
               residue_index=None):
    """StructureTokenEncoder.encode end to end (vqvae.py:291-322) + its kNN edges and the pre-quantisation z."""
    w = synth.vqenc_weights(d, VH, NL, 128, n_codes, seed=wseed)
    m = ref_import.build_vqenc(d, VH, NL, 128, n_codes, w)
    coords = synth.synth_backbone(B, L, seed=cseed, nan_frac=nan_frac, jitter=0.15)
    mask = torch.ones(B, L, dtype=torch.bool)
    if pad_from is not None:
        mask[-1, pad_from:] = False
        coords[-1, pad_from:] = float("nan")
    zs = []
    hook = m.pre_vq_proj.register_forward_hook(lambda _m, _i, o: zs.append(o.detach().float().clone()))
    with torch.no_grad():
        zq, idx = m.encode(coords, attention_mask=mask, sequence_id=sequence_id, residue_index=residue_index)
        from esm.utils.structure.affine3d import build_affine3d_from_coordinates
        _, amask = build_affine3d_from_coordinates(coords)
        sid = sequence_id if sequence_id is not None else torch.zeros(B, L, dtype=torch.int64)
        edges, emask = m.find_knn_edges(coords, ~mask, coord_mask=amask, sequence_id=sid, knn=16)
    arrays = dict(kind="vqenc", d=d, VH=VH, NL=NL, n_codes=n_codes, wseed=wseed, coords=_np(coords), attention_mask=_np(mask),
                  z_q=_np(zq), indices=_np(idx), z=_np(zs[0]), edges=_np(edges), edge_mask=_np(emask), affine_mask=_np(amask))
    if sequence_id is not None:
        arrays["sequence_id"] = _np(sequence_id)
    if residue_index is not None:
        arrays["residue_index"] = _np(residue_index)
    # No bf16 calibration here: the reference's encoder does not run in bf16 on the CPU, with .to(bfloat16) or under
    # autocast (its rotations are forced to fp32 and then meet bf16 projections: affine3d.py:168-175 raises).
    hook.remove()
    # margin of every token: gap between the nearest and the second nearest code (squared distances)
    flat = zs[0].reshape(-1, zs[0].shape[-1])
    cb = w["codebook.embeddings"]
    dist = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ cb.T + (cb.T ** 2).sum(0, keepdim=True)
    top2 = dist.topk(2, dim=1, largest=False).values
    arrays["margin"] = _np((top2[:, 1] - top2[:, 0]).reshape(B, L))
    print(name, "tokens", idx[0, :8].tolist(), "distinct", int(idx[amask].unique().numel()), "of", int(amask.sum()),
          "| z absmax", float(zs[0].abs().max()), "median margin", float(arrays["margin"][_np(amask)].mean()))
    save(name, **arrays)


def vqenc_cases():
    vqenc_case("vqenc_tiny", 128, 8, 2, 512, 2, 40, wseed=81, cseed=800, nan_frac=0.15, pad_from=31)
    sid = torch.zeros(2, 40, dtype=torch.int64)
    sid[0, 22:] = 1
    res = torch.arange(40)[None, :].repeat(2, 1)
    res[:, 20:] += 50  # a numbering gap: relative positions clamp at +-32
    vqenc_case("vqenc_tiny_seqid_resindex", 128, 8, 2, 512, 2, 40, wseed=81, cseed=801, sequence_id=sid, residue_index=res)
    vqenc_case("vqenc_tiny_short", 128, 8, 2, 512, 1, 10, wseed=81, cseed=802, nan_frac=0.0)  # L < knn
    vqenc_case("vqenc_tiny_sparse", 128, 8, 2, 512, 1, 48, wseed=81, cseed=803, nan_frac=0.8)  # few frames: sequence-distance edges
    vqenc_case("vqenc_d1024_vh128", 1024, 128, 2, 4096, 1, 64, wseed=82, cseed=804, nan_frac=0.1)


def fdec_case(name, d, H, NL, n_kw, n_ip, B, wseed, tseed, stride=1):
    """FunctionTokenDecoder.forward (function_decoder.py:127-176): [B, 8] function tokens -> three logit sets."""
    w = synth.fdec_weights(d, NL, n_kw, n_ip, seed=wseed)
    m = ref_import.build_fdec(d, H, NL, n_kw, n_ip, w)
    g = torch.Generator().manual_seed(tseed)
    tok = torch.randint(4, 260, (B, 8), generator=g)
    tok[1] = 0          # a <pad> row: every input becomes token 0
    tok[2, 3] = 2       # one special token inside a row zeroes the whole row too
    with torch.no_grad():
        out = m(tok)
    arrays = dict(kind="fdec", d=d, H=H, NL=NL, n_kw=n_kw, n_ip=n_ip, wseed=wseed, tokens=_np(tok), stride=stride)
    for k, v in out.items():
        arrays[k] = _np(v[:, ::stride])
        arrays["absmax_" + k] = float(v.abs().max())
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ob = m(tok)
    for k, v in out.items():
        arrays["ref_bf16_maxdiff_" + k] = float((ob[k].float() - v).abs().max())
        print(name, k, "absmax", arrays["absmax_" + k], "reference bf16 (autocast) drift", arrays["ref_bf16_maxdiff_" + k])
    save(name, **arrays)


def fdec_cases():
    fdec_case("fdec_tiny", 256, 2, 2, 300, 200, 9, wseed=91, tseed=900)
    fdec_case("fdec_d1024_3l", 1024, 8, 3, 58641, 29026, 6, wseed=92, tseed=901, stride=37)


def vqdec_cases():
    # tiny (2 blocks, d 128) incl. a bin-packed sequence_id, and the released geometry (30 blocks, d 1280, 20 heads)
    vqdec_case("vqdec_tiny", 128, 2, 2, 40, [40, 27], wseed=71, tseed=700)
    sid = torch.zeros(2, 40, dtype=torch.int64)
    sid[1, 27:] = 1
    vqdec_case("vqdec_tiny_seqid", 128, 2, 2, 40, [40, 27], wseed=71, tseed=701, sequence_id=sid)
    vqdec_case("vqdec_d1280_30l", 1280, 20, 30, 48, [48, 31], wseed=72, tseed=702)
    # the released geometry at the length the trunk is benchmarked at (262 144 residue pairs)
    vqdec_case("vqdec_d1280_30l_L512", 1280, 20, 30, 512, [512], wseed=72, tseed=703, pae_stride=4)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--text-only" in sys.argv:
        return text_cases()
    if "--sampling-only" in sys.argv:
        return sampling_cases()
    if "--600m-only" in sys.argv:
        esmc_case("esmc_300m_2x128", 960, 15, 30, [128, 101], wseed=31, tseed=300, store_hidden=(0, 14, 29), token_stride=2)
        return esmc_case("esmc_600m_2x128", 1152, 18, 36, [128, 77], wseed=32, tseed=310, store_hidden=(0, 17, 35),
                         token_stride=2)
    if "--vqdec-only" in sys.argv:
        return stack_case("stack_d1280_vqdec", 1280, 20, 30, 64, wseed=61, scale_residue=False)
    if "--vqdec-model-only" in sys.argv:
        return vqdec_cases()
    if "--vqenc-only" in sys.argv:
        return vqenc_cases()
    if "--fdec-only" in sys.argv:
        return fdec_cases()
    if "--stack-variants-only" in sys.argv:
        return stack_variant_cases()
    if "--fullsize-only" in sys.argv:
        return fullsize_cases()
    text_cases()
    sampling_cases()
    # (i) tiny ESMC under the three sequence_id dialects
    esmc_case("esmc_tiny_bool", 128, 2, 3, [33, 48, 17, 40], wseed=11, tseed=100)
    row0, sid0 = packed_row([20, 31, 9], 64)
    row1, sid1 = packed_row([40, 24], 64)
    tokens = torch.stack([row0, row1])
    w = synth.esmc_weights(128, 3, seed=11)
    m = ref_import.build_esmc(128, 2, 3, w)
    with torch.no_grad():
        out = m(tokens, sequence_id=torch.stack([sid0, sid1]))
    save("esmc_tiny_binpacked", kind="esmc", d=128, H=2, NL=3, wseed=11, tokens=_np(tokens), token_stride=1,
         sequence_id=_np(torch.stack([sid0, sid1])), sequence_logits=_np(out.sequence_logits),
         embeddings=_np(out.embeddings), hidden_layers=np.asarray([0, 2]),
         hidden_states=_np(out.hidden_states[[0, 2]]))
    # no-mask dialect: the stack with sequence_id=None (ESM3.logits / function decoder use this)
    wst = synth.make_weights(synth.trunk_shapes("", 128, 3), seed=12)
    st = ref_import.build_stack(128, 2, 3, wst)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 37, 128, generator=g)
    with torch.no_grad():
        post, pre, _ = st(x, sequence_id=None)
    save("stack_tiny_nomask", kind="stack", d=128, H=2, NL=3, wseed=12, x=_np(x), post=_np(post), pre=_np(pre),
         token_stride=1)
    # (ii) true-width trunks, 2 layers, L=64
    for d, H in ((960, 15), (1152, 18), (1536, 24), (2560, 40)):
        stack_case(f"stack_d{d}", d, H, 2, 64, wseed=20 + H)
    # the VQ-VAE structure decoder's stack: d1280, 20 heads, 30 layers, no residue scaling (vqvae.py:337-339)
    stack_case("stack_d1280_vqdec", 1280, 20, 30, 64, wseed=61, scale_residue=False)
    # (iii) full-depth ESMC-300M-shaped (C1 geometry), 2 x 128
    esmc_case("esmc_300m_2x128", 960, 15, 30, [128, 101], wseed=31, tseed=300, store_hidden=(0, 14, 29),
              token_stride=2)
    # the headline model shape at full depth (C2 geometry: d1152, 18 heads, 36 layers), 2 x 128
    esmc_case("esmc_600m_2x128", 1152, 18, 36, [128, 77], wseed=32, tseed=310, store_hidden=(0, 17, 35), token_stride=2)
    # (iv) geometric attention variants
    for v in ("plain", "nostruct", "multichain", "binpacked", "onerow_frameless"):
        geom_case(f"geom_{v}", 128, 8, 2, 48, wseed=41, variant=v)
    geom_case("geom_d1536_vh256", 1536, 256, 1, 40, wseed=43, variant="multichain")
    # (v) frames
    frames_case("frames")
    # (vi) ESM3 end to end, 2 layers
    esm3_case("esm3_tiny_alltracks", 128, 2, 8, 2, 2, 40, [40, 29], wseed=51, iseed=500)
    esm3_case("esm3_tiny_seqstruct", 128, 2, 8, 2, 1, 33, [33], wseed=51, iseed=501, with_tracks=False)
    chain = torch.zeros(2, 40, dtype=torch.int64)
    chain[0, 22:] = 1
    sid = torch.zeros(2, 40, dtype=torch.int64)
    sid[1, 15:] = 1
    esm3_case("esm3_tiny_chain_seqid", 128, 2, 8, 2, 2, 40, [40, 40], wseed=52, iseed=502,
              extra=dict(chain_id=chain, sequence_id=sid))
    esm3_case("esm3_d1536_2layer", 1536, 24, 256, 2, 1, 64, [64], wseed=53, iseed=503, with_tracks=False,
              big_stride=8)
    # (vii) VQ-VAE structure decoder, tokens -> coordinates / pLDDT / pTM / pae
    vqdec_cases()
    # (viii) VQ-VAE structure encoder, coordinates -> structure tokens
    vqenc_cases()
    # (ix) function-token decoder forward (block variant bias=True, qk_layernorm=False, GELU FFN, head dim 128)
    fdec_cases()
    # (x) the remaining block variants through the generic TransformerStack
    stack_variant_cases()
    # (xi) configs C3 / C5 at full sequence length
    fullsize_cases()


if __name__ == "__main__":
    main()
