"""World-size-2 gloo test of the data-parallel runner: sharding, the single all-gather, un-permutation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esm_amd.dp import run_data_parallel

WIDTH = 8


def fake_compute(tokens, lengths):
    """Deterministic per-residue 'embedding' that depends on the token and its position inside its sequence."""
    pos = torch.cat([torch.arange(n) for n in lengths]).to(tokens.device)
    base = tokens.float()[:, None] * 0.5 + pos.float()[:, None] * 0.25
    return base + torch.arange(WIDTH, dtype=torch.float32)[None, :]


def make_sequences():
    g = torch.Generator().manual_seed(11)
    lens = [5, 64, 17, 33, 2, 48, 9, 21, 40, 7, 3]
    return [torch.randint(4, 24, (n,), generator=g) for n in lens]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = make_sequences()
        res = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32)
        ok = all(torch.allclose(r, fake_compute(s, [s.numel()])) for r, s in zip(res, seqs))
        q.put((rank, ok, [tuple(r.shape) for r in res]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(120)
def test_dp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    seqs = make_sequences()
    for rank, ok, shapes in out:
        assert ok, f"rank {rank} got wrong results"
        assert shapes == [(s.numel(), WIDTH) for s in seqs]


def test_dp_single_process_matches_direct():
    seqs = make_sequences()
    res = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32)
    for r, s in zip(res, seqs):
        assert torch.allclose(r, fake_compute(s, [s.numel()]))
