"""Headline benchmark: residues/sec of the ESM C 600M forward (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config c2|c4]

N > 1 without WORLD_SIZE in the environment: this process only LAUNCHES - it starts N workers (one per GPU, plain child
processes, before anything here touches the GPU), waits for them and exits with the worst of their codes.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the workers are the launcher's.

--config c2 (default; BASELINE.json configs[1], the configuration the metric is quoted on)
    A step = one forward (embedding gather -> 36-block trunk -> sequence head) of one packed batch of 64 x 512
    synthetic residues per GPU, inputs resident in HBM.  Weak scaling: every rank runs its own batch with replicated
    weights and no data-path collective; the only collective is the result gather (bf16 per-residue embeddings,
    all_gather over RCCL on a side stream, overlapped with the next step).
--config c4 (BASELINE.json configs[3])
    A step = one pass over 4096 synthetic sequences with lengths log-uniform in [64, 2048] (seed 7), length-bucketed
    into packs of <= 32768 tokens and sharded over the ranks by esm_amd.dp (longest-processing-time-first on the FLOP
    model); per-residue bf16 embeddings are gathered to every rank round by round while the next pack computes
    (dp.run_data_parallel, streaming form).  Strong scaling: the job is fixed, ranks split it.

With more than one rank the gather is part of the step and MUST run over RCCL: a failing `nccl` initialisation or
pre-flight gather ends the run with a non-zero exit code and the reason (no silent fallback).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel = the FFN up-projection GEMM with the SwiGLU epilogue: algorithmic FLOPs / mean launch
                time from HIP events on the launch stream, against the 2.5 PFLOP/s dense bf16 MFMA peak
                (MI355X_MICROARCH.md).  `traffic` (HBM-side bytes per launch) is read from the PMC profile
                profiles/r03_pmc_bench.json when that profile was taken with THIS binary (esmk_build_id), else null.
  cpu_baseline  the CPU oracle (fp32 restatement of the reference, oracle/trunk_ref.py) timed on this host's
                cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D_MODEL, N_HEADS, N_LAYERS, FFN_H = 1152, 18, 36, 3072
BATCH, SEQ_LEN = 64, 512
PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X
PMC_PROFILE = os.path.join(ROOT, "profiles", "r03_pmc_bench.json")
C4_SEQS, C4_LMIN, C4_LMAX, C4_SEED, C4_BUDGET = 4096, 64, 2048, 7, 32768


def flop_per_residue(d=D_MODEL, h=FFN_H, L=SEQ_LEN, layers=N_LAYERS):
    """Algorithmic FLOPs (multiply-add = 2), SURVEY.md 8(d): per layer 8d^2 + 6dh + 4Ld, head 2d^2 + 2*d*64."""
    return layers * (8 * d * d + 6 * d * h + 4 * L * d) + 2 * d * d + 2 * d * 64


# ------------------------------------------------------------------------------------------------------------------
# launcher: runs in a process that never initialises the GPU
# ------------------------------------------------------------------------------------------------------------------
def self_launch(n: int) -> int:
    """Start n workers of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and wait."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ESM_AMD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:  # one worker failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


# ------------------------------------------------------------------------------------------------------------------
# measurement helpers (worker side)
# ------------------------------------------------------------------------------------------------------------------
def time_dominant_kernel(model, tokens, batch, steps=3):
    """Mean in-situ launch duration (s) of the FFN up-projection (SwiGLU) GEMM: HIP events recorded on the
    launch stream around that kernel in every layer of `steps` extra forwards (same launches rocprof sees)."""
    eng = model.transformer.engine()
    eng.enable_probes(True)
    ts = []
    for _ in range(steps):
        model.forward_packed(tokens, batch, want_hiddens=False, want_logits=True)
        ts += eng.probe_times_ms()
    eng.enable_probes(False)
    return sum(ts) / len(ts) * 1e-3


def measured_counters(build_id: str, workload: str):
    """PMC figures of the dominant kernel from profiles/r03_pmc_bench.json, only when that profile was taken with
    the binary that is running now and on this workload; otherwise (None, reason)."""
    try:
        prof = json.load(open(PMC_PROFILE))
    except (OSError, ValueError) as e:
        return None, f"no PMC profile ({type(e).__name__})"
    if prof.get("build_id") != build_id:
        return None, f"PMC profile is from build {prof.get('build_id')}, this binary is {build_id}"
    if prof.get("config") != workload:
        return None, f"PMC profile is for config {prof.get('config')}"
    return prof, "profiles/r03_pmc_bench.json (rocprofv3 --pmc, separate passes, this binary)"


def cpu_baseline(weights, n_seqs=6):
    """Oracle (port of the reference's fp32 path) on a bounded sample: n_seqs x 512 residues, all 36 layers;
    one warm-up pass, then the best of three."""
    import torch
    from esm_amd import synth
    from oracle import trunk_ref  # checker / baseline only
    tok = synth.synth_tokens([SEQ_LEN] * n_seqs, seed=4321)
    threads = torch.get_num_threads()
    times = []
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            trunk_ref.esmc_forward(tok, weights, N_HEADS, N_LAYERS)
            times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    return {"value": n_seqs * SEQ_LEN / dt, "unit": "residues/s", "cores": threads, "kind": "port",
            "sample": f"{n_seqs} x {SEQ_LEN} residues, ESM C 600M-shaped fp32, warm-up 1 + best of 3 passes "
                      f"({dt:.1f} s; all four {sum(times):.0f} s)"}


def c4_lengths():
    import numpy as np
    rng = np.random.default_rng(C4_SEED)
    return np.exp(rng.uniform(np.log(C4_LMIN), np.log(C4_LMAX), C4_SEQS)).astype(int).tolist()


def fail(msg: str):
    print(f"bench.py: {msg}", file=sys.stderr, flush=True)
    raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c2", "c4"], default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check without a GPU: workers form the gloo control plane, rank 0 prints a line")
    ap.add_argument("--rehearsal-shared-gpu", action="store_true",
                    help="allow more ranks than GPUs (1-GPU rehearsal): the gather then goes through host memory on "
                         "gloo and the line says so; never used for a reported number")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 10 if args.config == "c2" else 3
    if args.warmup is None:
        args.warmup = 3 if args.config == "c2" else 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))  # this process stays off the GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        fail(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import datetime

    import torch
    import torch.distributed as dist_mod

    dist = None
    if args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist_mod.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
            t = torch.tensor([float(rank)])
            dist_mod.all_reduce(t)
            assert t.item() == world * (world - 1) / 2
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "launcher": "self" if os.environ.get("ESM_AMD_BENCH_CHILD") else "external"}), flush=True)
        if world > 1:
            dist_mod.destroy_process_group()
        return

    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        fail("no GPU visible")
    shared = n_dev < world
    if shared and not args.rehearsal_shared_gpu:
        fail(f"{world} ranks but only {n_dev} GPU(s): one rank per GPU is required (RCCL cannot put two ranks on one "
             "device); --rehearsal-shared-gpu exists for 1-GPU rehearsals")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    if world > 1:
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, max-over-ranks of the elapsed time) on gloo: the timing contract must not depend
        # on the data-plane collective; RCCL ("nccl") carries the result gather.  No fallback: a failure ends the run.
        try:
            dist.init_process_group("gloo" if shared else "cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=300))
        except Exception as e:  # noqa: BLE001
            fail(f"init_process_group failed on rank {rank}: {type(e).__name__}: {e}")

    from esm_amd import _lib, synth
    from esm_amd.dp import esmc_compute_fn, run_data_parallel
    from esm_amd.engine import DeviceBatch
    from esm_amd.esmc import ESMC
    from esm_amd.packing import pack_lengths

    build_id = _lib.lib().esmk_build_id().decode()
    dev = torch.device("cuda", dev_index)
    weights = synth.esmc_weights(D_MODEL, N_LAYERS, seed=0)
    model = ESMC(D_MODEL, N_HEADS, N_LAYERS)
    model.load_state_dict(weights)
    model = model.to(dev)
    model._engines()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(torch.zeros(1))  # barrier on the gloo control plane (CPU tensor)
        torch.cuda.synchronize()

    n_tok = BATCH * SEQ_LEN
    tokens = synth.synth_tokens([SEQ_LEN] * BATCH, seed=1234 + 1000 * rank).reshape(-1).to(dev)
    batch = DeviceBatch(pack_lengths([SEQ_LEN] * BATCH), dev)

    if args.config == "c2":
        gather = world > 1
        side = torch.cuda.Stream() if gather else None
        if gather and shared:  # rehearsal only: host round trip on gloo
            send_h = torch.empty((n_tok, D_MODEL), dtype=torch.bfloat16).pin_memory()
            gathered_h = torch.empty((world * n_tok, D_MODEL), dtype=torch.bfloat16)
            gather_note = "REHEARSAL: ranks share a GPU, gather through host memory on gloo (not RCCL)"
        elif gather:
            gathered = torch.empty((world * n_tok, D_MODEL), dtype=torch.bfloat16, device=dev)
            send = torch.empty((n_tok, D_MODEL), dtype=torch.bfloat16, device=dev)
            gather_note = "all_gather_into_tensor of bf16 per-residue embeddings on a side stream (RCCL), overlapped with the next step"
        else:
            gather_note = "off (one rank)"

        def do_gather(post):
            if shared:
                send_h.copy_(post)  # synchronous device -> host copy with the bf16 cast
                dist.all_gather_into_tensor(gathered_h, send_h)
                return
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                send.copy_(post)  # fp32 -> bf16 wire format
                dist.all_gather_into_tensor(gathered, send)
            post.record_stream(side)

        def step():
            logits, post, _ = model.forward_packed(tokens, batch, want_hiddens=False, want_logits=True)
            if gather:
                do_gather(post)
            return logits

        if gather:  # pre-flight: one gather, fully synchronised and checked, before anything is timed
            try:
                _l, post0, _h = model.forward_packed(tokens, batch, want_hiddens=False, want_logits=False)
                do_gather(post0)
                if side is not None:
                    side.synchronize()
                torch.cuda.synchronize()
                got = (gathered_h if shared else gathered)[rank * n_tok:(rank + 1) * n_tok]
                if not torch.equal(got.cpu(), post0.to(torch.bfloat16).cpu()):
                    raise RuntimeError("gathered block of this rank differs from what it sent")
            except Exception as e:  # noqa: BLE001
                fail(f"pre-flight result gather failed on rank {rank}: {type(e).__name__}: {e}")

        def sync_step_streams():
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)

        residues_per_step = world * n_tok
        workload = ("ESM C 600M-shaped (d1152, 18 heads, 36 layers, random init) forward incl. sequence head, "
                    "64 x L=512 residues per GPU (BASELINE.json configs[1])")
        cfg = {"workload": workload, "global_batch": world * BATCH, "seq_len": SEQ_LEN, "parallelism": f"dp{world}",
               "result_gather": gather_note}
        scaling = "weak"
        fpr_total = residues_per_step * flop_per_residue()
    else:  # c4
        lengths = c4_lengths()
        seqs = [synth.synth_tokens([n], seed=50000 + i)[0] for i, n in enumerate(lengths)]
        compute = esmc_compute_fn(model)
        mode = "none" if world == 1 else "all"
        if shared:
            fail("--config c4 has no shared-GPU rehearsal mode")
        stats = {}
        sink = torch.zeros(D_MODEL, dtype=torch.float32, device=dev)

        def consume(idx, parts):  # stand-in for the caller's use of the results: touch every gathered byte once
            for t in parts:
                sink.add_(t.sum(dim=0, dtype=torch.float32))

        def step():
            run_data_parallel(seqs, compute, D_MODEL, d=D_MODEL, h=FFN_H, token_budget=C4_BUDGET, device=dev,
                              wire_dtype=torch.bfloat16, gather=mode, consume=consume, stats=stats)

        def sync_step_streams():
            pass

        if world > 1:
            try:
                run_data_parallel(seqs[:2 * world], compute, D_MODEL, d=D_MODEL, h=FFN_H, token_budget=C4_BUDGET, device=dev,
                                  wire_dtype=torch.bfloat16, gather="all", consume=consume)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                fail(f"pre-flight result gather failed on rank {rank}: {type(e).__name__}: {e}")
        residues_per_step = int(sum(lengths))
        workload = (f"ESM C 600M-shaped forward (embeddings, no head) of {C4_SEQS} synthetic sequences, L log-uniform in "
                    f"[{C4_LMIN}, {C4_LMAX}] (seed {C4_SEED}), packs <= {C4_BUDGET} tokens, sharded by esm_amd.dp "
                    "(BASELINE.json configs[3])")
        cfg = {"workload": workload, "sequences": C4_SEQS, "tokens_per_step": residues_per_step,
               "parallelism": f"dp{world}", "result_gather":
               "off (one rank)" if world == 1 else "all_gather_into_tensor per round on a side stream (RCCL), overlapped with the next pack"}
        scaling = "strong"
        fpr_total = float(sum(n * flop_per_residue(L=n) - n * (2 * D_MODEL * D_MODEL + 2 * D_MODEL * 64) for n in lengths))

    for _ in range(args.warmup):
        step()
    sync_step_streams()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_step_streams()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)  # CPU tensor -> gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = residues_per_step * args.steps / elapsed
        kt = time_dominant_kernel(model, tokens, batch)
        kflop = 2.0 * n_tok * (2 * FFN_H) * D_MODEL
        achieved = kflop / kt / 1e12
        prof, prof_note = measured_counters(build_id, "c2")
        line = {
            "metric": "residues/sec ESM C 600M fwd at L=512" if args.config == "c2" else "residues/sec ESM C 600M embedding, varlen L 64-2048",
            "value": value, "unit": "residues/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": cfg, "build_id": build_id,
            "mfma_frac_end_to_end": fpr_total * args.steps / elapsed / world / (PEAK_TFLOPS * 1e12),
            "flop_per_residue": flop_per_residue() if args.config == "c2" else fpr_total / residues_per_step,
            "roofline": {"bound": "mfma", "kernel": "gemm256_kernel<SWIGLU> (FFN up-projection 32768x6144x1152, measured on the 64 x 512 batch)",
                         "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                         "traffic": prof["dominant_kernel"]["hbm_bytes_per_launch"] if prof else None,
                         "traffic_unit": "bytes/launch", "traffic_source": prof_note,
                         "algorithmic_bytes": 2 * (n_tok * D_MODEL + 2 * FFN_H * D_MODEL + n_tok * FFN_H),
                         "launch_ms": kt * 1e3,
                         "mfma_busy_frac_at_held_clock": prof["dominant_kernel"].get("mfma_busy_frac") if prof else None},
        }
        if args.config == "c2":
            # the same batch through the reference's public entry point (ESMC.logits on a [64, 512] token matrix with
            # LogitsConfig(sequence, return_embeddings)): batch description on the device, output masking included
            from esm_amd.api import LogitsConfig, _BatchedESMProteinTensor
            inp = _BatchedESMProteinTensor(sequence=tokens.view(BATCH, SEQ_LEN))
            lcfg = LogitsConfig(sequence=True, return_embeddings=True)
            for _ in range(2):
                model.logits(inp, lcfg)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(5):
                model.logits(inp, lcfg)
            torch.cuda.synchronize()
            line["api_residues_per_s"] = 5 * n_tok / (time.perf_counter() - ta)
            line["api_note"] = "ESMC.logits(_BatchedESMProteinTensor[64, 512], LogitsConfig(sequence, return_embeddings)), 5 calls, one GPU"
            # the reference's DEFAULT forward also returns all 36 hidden states (esmc.py:167): [36, 64, 512, 1152] fp32 =
            # 5.4 GB per call, written by the residual-stream pass itself (no copies) and un-padded in place
            for _ in range(2):
                model(tokens.view(BATCH, SEQ_LEN))
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(5):
                model(tokens.view(BATCH, SEQ_LEN))
            torch.cuda.synchronize()
            line["api_hidden_states_residues_per_s"] = 5 * n_tok / (time.perf_counter() - ta)
            line["api_hidden_states_note"] = "ESMC.forward(tokens[64, 512]) with return_hidden_states=True (the reference's default), 5 calls"
        if args.config == "c4":
            line["dp"] = stats
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
