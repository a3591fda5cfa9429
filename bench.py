"""Headline benchmark: residues/sec of the ESM C 600M forward at L=512 (BASELINE.json config C2).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A step = one forward (embedding gather -> 36-block trunk -> sequence head) of one packed batch of
64 x 512 synthetic residues per GPU, inputs resident in HBM.  Weak scaling: every rank runs its own batch
with replicated weights and no data-path collective; the only collective is the result gather of the
data-parallel runner (bf16 per-residue embeddings, on a side stream, overlapped with the next step).
Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      dominant kernel = the FFN up-projection GEMM with the SwiGLU epilogue
                (32768 x 6144 x 1152): algorithmic FLOPs / mean launch time from HIP events on the launch stream,
                against the 2.5 PFLOP/s dense bf16 MFMA peak (MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (fp32 restatement of the reference, oracle/trunk_ref.py) timed on this host's
                cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

D_MODEL, N_HEADS, N_LAYERS, FFN_H = 1152, 18, 36, 3072
BATCH, SEQ_LEN = 64, 512
PEAK_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X
# HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 [gfx950 correction for
# 16-B/lane streams] + WRITE_SIZE, KiB -> bytes): profiles/r01_pmc_gemm_ffn1.json.  Algorithmic bytes: 291 MB.
DOMINANT_KERNEL_HBM_BYTES = 880_983_979
# Fraction of cycles the MFMA pipes are busy during that kernel, at the clock the chip held (rocprofv3 --pmc
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) over this very bench: tools/pmc_mfma_util.py,
# profiles/r01_pmc_mfma_util.json; whole run incl. weight packing: 0.434)
DOMINANT_KERNEL_MFMA_BUSY = 0.5993


def flop_per_residue(d=D_MODEL, h=FFN_H, L=SEQ_LEN, layers=N_LAYERS):
    """Algorithmic FLOPs (multiply-add = 2), SURVEY.md 8(d): per layer 8d^2 + 6dh + 4Ld, head 2d^2 + 2*d*64."""
    return layers * (8 * d * d + 6 * d * h + 4 * L * d) + 2 * d * d + 2 * d * 64


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


def cpu_baseline(weights, n_seqs=16):
    """Oracle (port of the reference's fp32 path) on a bounded sample: n_seqs x 512 residues, all 36 layers."""
    from esm_amd import synth
    from oracle import trunk_ref  # checker / baseline only
    tok = synth.synth_tokens([SEQ_LEN] * n_seqs, seed=4321)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    with torch.no_grad():
        trunk_ref.esmc_forward(tok, weights, N_HEADS, N_LAYERS)
    dt = time.perf_counter() - t0
    return {"value": n_seqs * SEQ_LEN / dt, "unit": "residues/s", "cores": threads, "kind": "port",
            "sample": f"{n_seqs} x {SEQ_LEN} residues, ESM C 600M-shaped fp32, one pass, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # one rank per GPU on a real node; ranks share a GPU only in rehearsals
    torch.cuda.set_device(dev_index)
    dist = None
    nccl_ok = False
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, max-over-ranks of the elapsed time) on gloo: the timing contract must not depend
        # on the data-plane collective; RCCL ("nccl") carries only the result all-gather.
        try:
            import datetime
            dist.init_process_group("cpu:gloo,cuda:nccl", timeout=datetime.timedelta(seconds=300))
            nccl_ok = n_dev >= world
        except Exception:
            dist.init_process_group("gloo")

    from esm_amd import synth
    from esm_amd.engine import DeviceBatch
    from esm_amd.esmc import ESMC
    from esm_amd.packing import pack_lengths

    dev = torch.device("cuda", dev_index)
    weights = synth.esmc_weights(D_MODEL, N_LAYERS, seed=0)
    model = ESMC(D_MODEL, N_HEADS, N_LAYERS)
    model.load_state_dict(weights)
    model = model.to(dev)
    model._engines()

    n_tok = BATCH * SEQ_LEN
    tokens = synth.synth_tokens([SEQ_LEN] * BATCH, seed=1234 + 1000 * rank).reshape(-1).to(dev)
    batch = DeviceBatch(pack_lengths([SEQ_LEN] * BATCH), dev)

    gather = (world > 1) and nccl_ok and not args.no_gather
    side = torch.cuda.Stream() if gather else None
    gathered = torch.empty((world * n_tok, D_MODEL), dtype=torch.bfloat16, device=dev) if gather else None
    send = torch.empty((n_tok, D_MODEL), dtype=torch.bfloat16, device=dev) if gather else None
    gather_note = "off" if (world == 1 or args.no_gather) else ("off: fewer GPUs than ranks" if not nccl_ok else "pending")

    def do_gather(post):
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

    if gather:  # pre-flight: one gather, fully synchronised, before anything is timed; failure -> measure without it
        try:
            _l, post0, _h = model.forward_packed(tokens, batch, want_hiddens=False, want_logits=False)
            do_gather(post0)
            side.synchronize()
            torch.cuda.synchronize()
            gather_note = "all_gather_into_tensor of bf16 per-residue embeddings on a side stream (RCCL), overlapped with the next step"
        except Exception as e:
            gather = False
            gather_note = f"failed in pre-flight: {type(e).__name__}: {e}"[:200]
    if dist is not None:  # every rank must agree on whether the collective is part of the step
        flag = torch.tensor([1.0 if gather else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if gather and flag.item() == 0.0:
            gather = False
            gather_note = "off: pre-flight failed on another rank"

    def sync_all():
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if dist is not None:
            dist.all_reduce(torch.zeros(1))  # barrier on the gloo control plane (CPU tensor)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)  # CPU tensor -> gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * n_tok * args.steps / elapsed
        fpr = flop_per_residue()
        kt = time_dominant_kernel(model, tokens, batch)
        kflop = 2.0 * n_tok * (2 * FFN_H) * D_MODEL
        achieved = kflop / kt / 1e12
        line = {
            "metric": "residues/sec ESM C 600M fwd at L=512", "value": value, "unit": "residues/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ESM C 600M-shaped (d1152, 18 heads, 36 layers, random init) forward incl. "
                                   "sequence head, 64 x L=512 residues per GPU (BASELINE.json configs[1])",
                       "global_batch": world * BATCH, "seq_len": SEQ_LEN, "parallelism": f"dp{world}",
                       "result_gather": gather_note},
            "mfma_frac_end_to_end": value / world * fpr / (PEAK_TFLOPS * 1e12),
            "flop_per_residue": fpr,
            "roofline": {"bound": "mfma", "kernel": "gemm256_kernel<SWIGLU> (FFN up-projection 32768x6144x1152)",
                         "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                         "traffic": DOMINANT_KERNEL_HBM_BYTES, "traffic_unit": "bytes/launch (PMC, profiles/r01_pmc_gemm_ffn1.json)",
                         "launch_ms": kt * 1e3,
                         "mfma_busy_frac_at_held_clock": DOMINANT_KERNEL_MFMA_BUSY,
                         "mfma_busy_source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE "
                                             "(profiles/r01_pmc_mfma_util.json)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.all_reduce(torch.zeros(1))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
