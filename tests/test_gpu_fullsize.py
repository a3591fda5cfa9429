"""BASELINE.json configs C3 and C5 at their FULL size on the HIP path (round-1 verdict: these two were only tested
at 2 layers x L=64).

C3  ESM3-open-shaped (d1536, 24 heads, 256 geometric heads, 48 layers) on 16 x 512 residues in one forward:
    row 0 is the input of tests/golden/esm3_open_48l_L512.npz and is checked against the IMPORTED REFERENCE's outputs
    stored there (all six heads + embeddings); two more rows are checked against the CPU oracle run on those rows.
C5  6B-shaped trunk (d2560, 40 heads): two layers at L = 4096 against tests/golden/stack_d2560_L4096.npz (reference
    TransformerStack) and against the oracle; then the 80-layer 4 x 4096 forward for finiteness, determinism and row
    independence (the reference has no 6B builder and a CPU oracle pass of 80 x 4 x 4096 takes minutes, so the full
    depth is covered by properties, the arithmetic by the 2-layer fixture at the full sequence length).

Gate (north_star): per-residue cosine >= 0.999 on valid positions, logits within 0.05 + 0.02 max|logit_ref|.
"""
import pytest
import torch

from esm_amd import synth
from tests import _golden as G

pytestmark = pytest.mark.gpu


def logits_ok(got, ref, what=""):
    lim = 0.05 + 0.02 * ref.abs().max().item()
    err = (got - ref).abs().max().item()
    assert err <= lim, (what, err, lim)


def emb_ok(got, ref, what="", cos_min=0.999):
    c = G.cosine_min(got, ref)
    assert c >= cos_min, (what, c)
    rel = (got - ref).norm().item() / max(ref.norm().item(), 1e-30)
    assert rel <= 0.03, (what, rel)


def test_c3_esm3_open_48_layers_16x512():
    from esm_amd.esm3 import ESM3
    from oracle import trunk_ref  # checker only
    fx = G.load("esm3_open_48l_L512")
    d, H, VH, NL = int(fx["d"]), int(fx["H"]), int(fx["VH"]), int(fx["NL"])
    assert (d, H, VH, NL) == (1536, 24, 256, 48)
    w = G.weights_for(fx)
    m = ESM3(d, H, VH, NL)
    m.load_state_dict(w)
    m = m.cuda()
    B, L = 16, 512
    kw0 = G.esm3_kwargs(fx)  # row 0: sequence + structure tokens + coordinates of the fixture
    g = torch.Generator().manual_seed(77)
    seq = synth.synth_tokens([L] * B, seed=2024)
    struct = torch.randint(0, 4096, (B, L), generator=g)
    coords = synth.synth_backbone(B, L, seed=31, nan_frac=0.1)
    coords[:, 0] = float("inf")
    coords[:, L - 1] = float("inf")
    seq[0], struct[0], coords[0] = kw0["sequence_tokens"][0], kw0["structure_tokens"][0], kw0["structure_coords"][0]
    chain = torch.zeros(B, L, dtype=torch.int64)
    chain[5, 300:] = 1  # one two-chain row (hard mask in the geometric block); row 0 keeps the fixture's default
    out = m(sequence_tokens=seq.cuda(), structure_tokens=struct.cuda(), structure_coords=coords.cuda(),
            chain_id=chain.cuda())
    torch.cuda.synchronize()
    for k in ("sequence_logits", "structure_logits", "secondary_structure_logits", "sasa_logits", "function_logits",
              "residue_logits", "embeddings"):
        assert torch.isfinite(getattr(out, k)).all(), k
    # --- row 0 against the reference's own outputs (golden fixture)
    es, bs = int(fx["emb_stride"]), int(fx["big_stride"])
    emb_ok(out.embeddings[0:1, ::es].cpu(), G.t(fx["embeddings"]), "embeddings vs reference")
    for k in ("sequence_logits", "secondary_structure_logits", "sasa_logits"):
        logits_ok(getattr(out, k)[0:1].cpu(), G.t(fx[k]), k)
    for k in ("structure_logits", "function_logits", "residue_logits"):
        logits_ok(getattr(out, k)[0:1, ::bs].cpu(), G.t(fx[k]), k)
    # --- two more rows of the same forward against the oracle run on those rows (the two-chain row included)
    rows = [5, 15]
    ref = trunk_ref.esm3_forward(w, H, VH, NL, sequence_tokens=seq[rows], structure_tokens=struct[rows],
                                 structure_coords=coords[rows], chain_id=chain[rows])
    emb_ok(out.embeddings[rows].cpu(), ref["embeddings"], "embeddings vs oracle")
    for k in ("sequence_logits", "structure_logits", "secondary_structure_logits", "sasa_logits", "function_logits",
              "residue_logits"):
        logits_ok(getattr(out, k)[rows].cpu(), ref[k], k + " vs oracle")


def test_c5_d2560_two_layers_L4096_vs_reference_and_oracle():
    from esm_amd.transformer_stack import TransformerStack
    from oracle import trunk_ref  # checker only
    fx = G.load("stack_d2560_L4096")
    d, H, NL = int(fx["d"]), int(fx["H"]), int(fx["NL"])
    assert (d, H) == (2560, 40)
    w = G.weights_for(fx)
    m = TransformerStack(d, H, None, NL, n_layers_geom=0)
    m.load_state_dict(w)
    m = m.cuda()
    x = synth.synth_activations(tuple(int(v) for v in fx["x_shape"]), seed=int(fx["x_seed"]))
    assert x.shape[1] == 4096
    sid = G.t(fx["sequence_id"])
    post, pre, hid = m(x.cuda(), sid.cuda())
    torch.cuda.synchronize()
    s = int(fx["token_stride"])
    valid = sid[:, ::s]
    emb_ok(post.cpu()[:, ::s][valid], G.t(fx["post"])[valid], "post vs reference")
    emb_ok(pre.cpu()[:, ::s][valid], G.t(fx["pre"])[valid], "pre vs reference")
    # every valid position (not only the stored stride) against the oracle
    rpost, rpre, _ = trunk_ref.transformer_stack(x, sid, w, "", H, NL)
    emb_ok(post.cpu()[sid], rpost[sid], "post vs oracle")
    emb_ok(pre.cpu()[sid], rpre[sid], "pre vs oracle")


def test_c5_80_layers_4x4096_properties():
    """The C5 bench workload itself: finite, deterministic, and every row depends on its own tokens only."""
    from esm_amd.engine import DeviceBatch
    from esm_amd.packing import pack_lengths
    from esm_amd.transformer_stack import TransformerStack
    d, H, NL, B, L = 2560, 40, 80, 4, 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    wd = {}
    for k, shp in synth.trunk_shapes("", d, NL).items():  # drawn on the device: 6.3 G parameters
        mean, std = synth._kind(k)
        wd[k] = torch.randn(shp, device="cuda", generator=g) * std + mean
    m = TransformerStack(d, H, None, NL, n_layers_geom=0)
    m.load_state_dict(wd)
    del wd
    m = m.cuda()
    eng = m.engine()
    x0 = torch.randn(B * L, d, device="cuda", generator=g)
    batch = DeviceBatch(pack_lengths([L] * B), "cuda")
    post_a, pre_a, _ = eng.forward(x0.clone(), batch, want_post=True, want_hiddens=False)
    post_b, pre_b, _ = eng.forward(x0.clone(), batch, want_post=True, want_hiddens=False)
    torch.cuda.synchronize()
    assert torch.isfinite(post_a).all() and torch.isfinite(pre_a).all()
    assert pre_a.std(dim=-1).min().item() > 0.1  # rows of the residual stream are not degenerate
    assert torch.equal(post_a, post_b) and torch.equal(pre_a, pre_b)
    # row 2 alone: another batch shape (other GEMM tile schedule, same arithmetic up to fp32 summation order)
    one = DeviceBatch(pack_lengths([L]), "cuda")
    post_1, _pre_1, _ = eng.forward(x0[2 * L:3 * L].clone(), one, want_post=True, want_hiddens=False)
    c = torch.nn.functional.cosine_similarity(post_1, post_a[2 * L:3 * L], dim=-1).min().item()
    assert c >= 0.999, c
    # and it must NOT equal another row
    c_other = torch.nn.functional.cosine_similarity(post_1, post_a[:L], dim=-1).mean().item()
    assert c_other < 0.9, c_other
