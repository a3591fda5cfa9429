"""Host logic of the sampling loop (CPU): step masks and annealing against fixtures produced by the reference's
generation.py / sampling.py (oracle/make_golden.py sampling_cases).  The sampling mask and the SASA decoding are device
kernels: tests/test_gpu_sampling.py."""
import os

import numpy as np
import torch

from esm_amd.sampling import GenerationConfig, annealed_temperature, choose_positions, step_unmask_count

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling.npz"))


def test_annealed_temperature_matches_reference():
    for row, (n, T) in zip(G["annealed"], ((8, 0.7), (8, 1.5))):
        got = [annealed_temperature(t, n, T) for t in range(n)]
        np.testing.assert_allclose(got, row, rtol=0, atol=0)
    assert annealed_temperature(0, 1, 1.0) == float(G["annealed_n1"])


def test_step_masks_match_reference():
    for i in range(int(G["n_step_mask"])):
        k = f"sm{i}_"
        tokens = torch.from_numpy(G[k + "tokens"])
        ent = torch.from_numpy(G[k + "entropy"])
        mask_idx, num_steps = int(G[k + "mask_idx"]), int(G[k + "num_steps"])
        L = tokens.shape[1]
        cur = tokens.clone()
        inside = torch.ones(1, L, dtype=torch.bool)
        inside[:, 0] = False
        inside[:, -1] = False
        total = torch.tensor([int(G[k + "n_masked"])])
        for t in range(num_steps):
            cand = (cur == mask_idx) & inside
            n = step_unmask_count(t, num_steps, str(G[k + "schedule"]), total, cand.sum(-1))
            where = choose_positions(cand, n, "entropy", ent)
            assert np.array_equal(where.numpy(), G[k + "where"][t]), (i, t)
            cur = torch.where(where, torch.full_like(cur, 5), cur)
        assert not (cur == mask_idx).any()  # everything decoded after the last step


def test_choose_positions_random_counts():
    g = torch.Generator().manual_seed(0)
    cand = torch.zeros(3, 30, dtype=torch.bool)
    cand[0, 2:20] = True
    cand[1, 5:8] = True
    n = torch.tensor([7, 3, 4])
    where = choose_positions(cand, n, "random", None, generator=g)
    assert where.sum(-1).tolist() == [7, 3, 0] and not (where & ~cand).any()


def test_generation_config_validation():
    import pytest
    with pytest.raises(ValueError):
        GenerationConfig(track="sequence", schedule="nope")
    with pytest.raises(ValueError):
        GenerationConfig(track="sequence", strategy="nope")
    c = GenerationConfig(track="sequence")
    c.use_entropy_based_unmasking_strategy()
    assert (c.schedule, c.strategy, c.temperature_annealing) == ("cosine", "entropy", False)
