"""Shared helpers: load a golden fixture, rebuild its weights, run the oracle on its inputs."""
import os

import numpy as np
import torch

from esm_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def seeded(fx, key):
    """Input tensor `key` of a fixture: stored, or (large inputs) regenerated from the seed the fixture names."""
    if key in fx:
        return t(fx[key])
    return synth.synth_activations(tuple(int(v) for v in fx[key + "_shape"]), seed=int(fx[key + "_seed"]))


def esm3_kwargs(fx):
    return {k[3:]: t(v) for k, v in fx.items() if k.startswith("in_")}


def weights_for(fx):
    kind = str(fx["kind"])
    d, ws = int(fx["d"]), int(fx["wseed"])
    if kind == "esmc":
        return synth.esmc_weights(d, int(fx["NL"]), seed=ws)
    if kind == "stack":
        return synth.make_weights(synth.trunk_shapes("", d, int(fx["NL"])), seed=ws)
    if kind == "esm3":
        return synth.esm3_weights(d, int(fx["NL"]), int(fx["VH"]), seed=ws)
    if kind == "geom":
        shapes = {k: v for k, v in synth.trunk_shapes("", d, 1, v_heads=int(fx["VH"]), n_layers_geom=1).items()
                  if "geom_attn" in k}
        return synth.make_weights(shapes, seed=ws)
    if kind == "vqdec":
        return synth.make_weights(synth.vqdec_shapes(d, int(fx["NL"])), seed=ws)
    if kind == "vqenc":
        return synth.vqenc_weights(d, int(fx["VH"]), int(fx["NL"]), 128, int(fx["n_codes"]), seed=ws)
    if kind == "fdec":
        return synth.fdec_weights(d, int(fx["NL"]), int(fx["n_kw"]), int(fx["n_ip"]), seed=ws)
    if kind == "stack_variant":
        # shapes from OUR module's state_dict: the fixture's weights were drawn over the REFERENCE module's keys
        # and shapes, so this also pins the state_dict contract of every block variant
        from esm_amd.transformer_stack import TransformerStack
        m = TransformerStack(d, int(fx["H"]), int(fx["VH"]) or None, int(fx["NL"]), **variant_ctor(fx))
        return synth.stack_variant_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=ws)
    raise ValueError(kind)


def variant_ctor(fx) -> dict:
    """Constructor options of a stack_variant fixture (stored as repr(sorted(kwargs.items())))."""
    import ast
    return dict(ast.literal_eval(str(fx["ctor"])))


def cosine_min(a, b):
    """Minimum per-row cosine similarity over the last dim."""
    a = a.double().reshape(-1, a.shape[-1])
    b = b.double().reshape(-1, b.shape[-1])
    num = (a * b).sum(-1)
    den = a.norm(dim=-1) * b.norm(dim=-1)
    return float((num / den.clamp_min(1e-30)).min())


def assert_knn_equal(edges, edge_mask, fx):
    """kNN edges against a vqenc fixture.  Exact where the reference's order is determined: for a residue WITH
    coordinates, the leading neighbours that have coordinates too (true CA distances, all distinct in the fixtures).
    Behind them come residues ranked by sequence distance, where i-k and i+k tie and the reference's unstable
    torch.sort leaves the order to the platform; there (and for residues without coordinates, whose output is zeroed)
    only the finite-distance mask and the multiset of sequence distances must agree."""
    ref_e, ref_m = t(fx["edges"]), t(fx["edge_mask"])
    amask = t(fx["affine_mask"])
    assert edges.shape == ref_e.shape and torch.equal(edge_mask.bool(), ref_m)
    B, L, E = ref_e.shape
    for b in range(B):
        for i in range(L):
            got, want = edges[b, i], ref_e[b, i]
            if amask[b, i]:
                n_lead = 0
                while n_lead < E and ref_m[b, i, n_lead] and amask[b, want[n_lead]]:
                    n_lead += 1
                assert torch.equal(got[:n_lead], want[:n_lead]), (b, i)
                assert int(got[0]) == i
            fin = ref_m[b, i]
            assert torch.equal((got[fin] - i).abs().sort().values, (want[fin] - i).abs().sort().values), (b, i)
