"""Host-side logic: sequence_id dialects -> segments, attention plan, DP bucketing, module key compatibility."""
import numpy as np
import pytest
import torch

from esm_amd import synth
from esm_amd.dp import assign_packs, make_packs, plan, sequence_cost
from esm_amd.packing import pack_lengths, pack_sequence_id, plan_attention


def dense_mask_from_packed(pb):
    """[N, N] attend-mask implied by cu_seqlens, for comparison with the reference rule."""
    n = pb.n_tokens
    m = np.zeros((n, n), dtype=bool)
    for a, b in zip(pb.cu_seqlens[:-1], pb.cu_seqlens[1:]):
        m[a:b, a:b] = True
    return m


def reference_mask(sid):
    """esm/layers/attention.py:67 per row, block-diagonal over rows."""
    B, L = sid.shape
    m = np.zeros((B * L, B * L), dtype=bool)
    for b in range(B):
        m[b * L:(b + 1) * L, b * L:(b + 1) * L] = sid[b][:, None] == sid[b][None, :]
    return m


def test_none_dialect():
    pb = pack_sequence_id(None, 3, 7)
    assert pb.n_tokens == 21 and pb.n_segments == 3 and pb.gather is None
    assert np.array_equal(pb.cu_seqlens, [0, 7, 14, 21])
    assert np.array_equal(pb.positions[:8], [0, 1, 2, 3, 4, 5, 6, 0])


def test_int_dialect_matches_reference_mask():
    sid = np.array([[0, 0, 0, 1, 1, 2, 2, 2], [5, 5, 5, 5, 5, 5, 9, 9]])
    pb = pack_sequence_id(sid, 2, 8)
    assert np.array_equal(dense_mask_from_packed(pb), reference_mask(sid))
    assert np.array_equal(pb.positions, np.tile(np.arange(8), 2))  # column index, does not restart (rotary.py:207)
    assert np.array_equal(pb.row_cu, [0, 8, 16])


def test_bool_dialect_without_drop_keeps_pad_segment():
    sid = np.array([[True, True, True, False, False]])
    pb = pack_sequence_id(sid, 1, 5, drop_false=False)
    assert np.array_equal(dense_mask_from_packed(pb), reference_mask(sid))  # pads attend to pads (SDPA path)


def test_bool_dialect_drop():
    sid = np.array([[True, True, True, False, False], [True, True, True, True, True], [False] * 5])
    pb = pack_sequence_id(sid, 3, 5, drop_false=True)
    assert pb.n_tokens == 8 and np.array_equal(pb.cu_seqlens, [0, 3, 8])
    assert np.array_equal(pb.gather, [0, 1, 2, 5, 6, 7, 8, 9])
    assert np.array_equal(pb.positions, [0, 1, 2, 0, 1, 2, 3, 4])
    assert np.array_equal(pb.row_cu, [0, 3, 8, 8])


def test_non_contiguous_ids_sorted_into_segments():
    """An id that re-appears in a non-adjacent run: the row is stably sorted by id; `gather` is the permutation and
    positions keep the original columns (attention.py:67 is an equality mask, order inside a segment is irrelevant)."""
    pb = pack_sequence_id(np.array([[0, 0, 1, 0], [5, 5, 5, 5]]), 2, 4)
    assert np.array_equal(pb.gather, [0, 1, 3, 2, 4, 5, 6, 7])
    assert np.array_equal(pb.positions, [0, 1, 3, 2, 0, 1, 2, 3])
    assert np.array_equal(pb.cu_seqlens, [0, 3, 4, 8])
    assert np.array_equal(pb.seq_id, [0, 0, 0, 1, 5, 5, 5, 5])
    assert np.array_equal(pb.row_cu, [0, 4, 8])
    # contiguous ids stay untouched
    assert pack_sequence_id(np.array([[0, 0, 1, 1]]), 1, 4).gather is None


def test_shape_mismatch():
    with pytest.raises(ValueError):
        pack_sequence_id(np.zeros((2, 3), dtype=np.int64), 2, 4)


def test_plan_attention():
    m = plan_attention(np.array([0, 1, 130, 130 + 256], dtype=np.int32))
    assert m.tolist() == [[0, 0], [1, 0], [1, 128], [2, 0], [2, 128]]
    assert pack_lengths([3, 200]).qblk_map.tolist() == [[0, 0], [1, 0], [1, 128]]
    with pytest.raises(ValueError):
        pack_lengths([3, 0])


def test_packs_respect_budget_and_cover_everything():
    rng = np.random.default_rng(7)
    lengths = np.exp(rng.uniform(np.log(64), np.log(2048), 500)).astype(int)
    packs = make_packs(lengths, 32768)
    flat = sorted(i for p in packs for i in p)
    assert flat == list(range(500))
    assert all(sum(lengths[i] for i in p) <= 32768 for p in packs)
    # similar lengths inside a pack (sorted fill)
    for p in packs:
        ls = [lengths[i] for i in p]
        assert ls == sorted(ls, reverse=True)
    with pytest.raises(ValueError):
        make_packs([10, 40000], 32768)


def test_lpt_assignment_balances_cost():
    rng = np.random.default_rng(3)
    lengths = np.exp(rng.uniform(np.log(64), np.log(2048), 4096)).astype(int)
    packs, per_rank, costs = plan(lengths, 8, 32768, 1152, 3072)
    loads = [sum(costs[p] for p in r) for r in per_rank]
    assert max(loads) / (sum(loads) / 8) < 1.05
    assert sorted(p for r in per_rank for p in r) == list(range(len(packs)))
    assert sequence_cost(512, 1152, 3072) == pytest.approx(512 * (8 * 1152**2 + 6 * 1152 * 3072) + 4 * 1152 * 512**2)


def test_module_state_dict_keys_match_reference_contract():
    from esm_amd.esm3 import ESM3
    from esm_amd.esmc import ESMC
    m = ESMC(128, 2, 2)
    assert set(m.state_dict().keys()) == set(synth.esmc_shapes(128, 2).keys())
    m3 = ESM3(128, 2, 8, 2)
    assert set(m3.state_dict().keys()) == set(synth.esm3_shapes(128, 2, 8).keys())
    for k, shp in synth.esm3_shapes(128, 2, 8).items():
        assert tuple(m3.state_dict()[k].shape) == tuple(shp), k


def test_cpu_module_fails_loudly_without_gpu_path():
    from esm_amd._lib import EsmkError
    from esm_amd.esmc import ESMC
    m = ESMC(128, 2, 1)
    with pytest.raises(EsmkError):
        m(torch.zeros(1, 4, dtype=torch.long))


def test_block_variants_state_dict_and_rejections():
    """Every constructor option of the reference's TransformerStack builds the reference's state_dict keys (the
    fixtures' weights were drawn over the REFERENCE module's keys / shapes: tests/_golden.py weights_for); the one
    combination that is not built is rejected at construction."""
    from esm_amd.transformer_stack import TransformerStack
    from tests import _golden as G
    for name in ("stackvar_bias_noqkln_gelu", "stackvar_noqkln_swiglu", "stackvar_gelu_noscale", "stackvar_geom_nozero",
                 "stackvar_bias_geom2"):
        fx = G.load(name)
        m = TransformerStack(int(fx["d"]), int(fx["H"]), int(fx["VH"]) or None, int(fx["NL"]), **G.variant_ctor(fx))
        m.load_state_dict(G.weights_for(fx), strict=True)
        assert not m.fused
    m = TransformerStack(128, 2, None, 1, n_layers_geom=0, bias=True, qk_layernorm=False, ffn_type="gelu", expansion_ratio=4.0)
    keys = set(m.state_dict())
    assert "blocks.0.attn.out_proj.bias" in keys and "blocks.0.ffn.1.bias" in keys and "blocks.0.attn.q_ln.weight" not in keys
    assert m.state_dict()["blocks.0.ffn.1.weight"].shape == (512, 128)
    assert TransformerStack(128, 2, 8, 2, n_layers_geom=1, mask_and_zero_frameless=True).fused
    assert TransformerStack(128, 2, None, 2, n_layers_geom=0).fused
    biased = TransformerStack(128, 2, None, 1, n_layers_geom=0, bias=True)  # biased q_ln / k_ln (attention.py:32-33)
    assert "blocks.0.attn.q_ln.bias" in biased.state_dict() and not biased.fused
    with pytest.raises(ValueError):
        TransformerStack(128, 2, None, 1, n_layers_geom=0, ffn_type="relu")
    with pytest.raises(ValueError):
        TransformerStack(128, 2, None, 1, n_layers_geom=1)  # v_heads missing


def test_batched_protein_tensor_slice_roundtrip():
    from esm_amd.api import ESMProteinTensor, _BatchedESMProteinTensor
    b = _BatchedESMProteinTensor(sequence=torch.arange(12).view(2, 6), coordinates=torch.zeros(2, 6, 3, 3))
    assert len(b) == 6 and b.batch_size == 2
    r = b.slice(1, sequence_len=4)
    assert isinstance(r, ESMProteinTensor) and r.sequence.tolist() == [6, 7, 8, 9] and r.coordinates.shape == (4, 3, 3)
    assert r.structure is None
    b.set_slice(0, ESMProteinTensor(sequence=torch.full((6,), 9)))
    assert b.sequence[0].tolist() == [9] * 6 and b.sequence[1].tolist() == [6, 7, 8, 9, 10, 11]
    with pytest.raises(AssertionError):
        b.set_slice(0, ESMProteinTensor(structure=torch.zeros(6, dtype=torch.long)))
