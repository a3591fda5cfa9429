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
    lens = [5, 64, 17, 33, 2, 48, 9, 21, 40, 7, 3, 60, 50, 30, 64, 12]   # best-fit packs of <= 64 tokens: 8 packs, 4 rounds on 2 ranks
    return [torch.randint(4, 24, (n,), generator=g) for n in lens]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = make_sequences()
        want = [fake_compute(s, [s.numel()]) for s in seqs]
        res = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32)
        ok = all(torch.allclose(r, w) for r, w in zip(res, want))
        # gather to rank 0 only
        root = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32,
                                 gather="root")
        if rank == 0:
            ok = ok and all(torch.allclose(r, w) for r, w in zip(root, want))
        else:
            ok = ok and all(r is None for r in root)
        # streaming form: results are handed to a callback round by round and nothing is retained; at most two
        # rounds of results are resident at any time although this plan has several rounds per rank
        seen, st = {}, {}

        def consume(idx, parts):
            for i, t in zip(idx, parts):
                assert i not in seen
                seen[i] = t.sum(dim=0)  # a pooled quantity: the views themselves are not kept

        out = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=64, wire_dtype=torch.float32,
                                consume=consume, stats=st)
        ok = ok and out is None and sorted(seen) == list(range(len(seqs)))
        ok = ok and all(torch.allclose(seen[i], want[i].sum(dim=0)) for i in seen)
        ok = ok and st["rounds"] >= 3 and st["peak_resident_rounds"] <= 2 and st["collectives"] == st["rounds"]
        # no collective: a rank keeps its own sequences only
        own = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32,
                                gather="none")
        mine = [i for i, r in enumerate(own) if r is not None]
        ok = ok and 0 < len(mine) < len(seqs) and all(torch.allclose(own[i], want[i]) for i in mine)
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


def _subgroup_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group([1, 2])  # a subgroup WITHOUT global rank 0: its root is global rank 1
        ok = True
        if rank in (1, 2):
            seqs = make_sequences()
            want = [fake_compute(s, [s.numel()]) for s in seqs]
            for mode in ("root", "all"):
                res = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32,
                                        gather=mode, group=grp)
                if mode == "all" or rank == 1:
                    ok = ok and all(torch.allclose(r, w) for r, w in zip(res, want))
                else:
                    ok = ok and all(r is None for r in res)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dp_gather_root_inside_a_subgroup():
    """gather="root" with a process group that does not contain global rank 0 (advisor, round 2): the runner's root is
    the group's rank 0, and dist.gather must be given its GLOBAL rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(ok for _rank, ok in out), out


def test_dp_single_process_matches_direct():
    seqs = make_sequences()
    res = run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32)
    for r, s in zip(res, seqs):
        assert torch.allclose(r, fake_compute(s, [s.numel()]))
    st = {}
    run_data_parallel(seqs, fake_compute, WIDTH, d=64, h=256, token_budget=80, wire_dtype=torch.float32,
                      consume=lambda idx, parts: None, stats=st)
    assert st["collectives"] == 0 and st["peak_resident_rounds"] <= 2


@pytest.mark.timeout(180)
def test_bench_self_launcher_two_workers():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its own two workers (the parent never touches a GPU),
    they rendezvous on 127.0.0.1 and rank 0 prints one JSON line; --dry-run keeps the GPU out of it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "launcher": "self"}
    # a worker that fails makes the launcher exit non-zero (here: --gpus disagrees with the forced WORLD_SIZE)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
