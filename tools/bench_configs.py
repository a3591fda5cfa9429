"""Throughput of the other BASELINE.json configs (C1, C3, C4, C5) on one GPU.  Random weights are drawn on the
device with torch (performance does not depend on values); parity for these shapes is covered by tests/.
   python tools/bench_configs.py [c1] [c3] [c4] [c5]"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from esm_amd import synth
from esm_amd.dp import make_packs
from esm_amd.engine import DeviceBatch
from esm_amd.packing import pack_lengths

PEAK = 2500e12


def dev_weights(shapes):
    g = torch.Generator(device="cuda").manual_seed(0)
    w = {}
    for k, shp in shapes.items():
        mean, std = synth._kind(k)
        w[k] = torch.randn(shp, device="cuda", generator=g) * std + mean
    return w


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def flop_per_token(d, h, layers, L):
    return layers * (8 * d * d + 6 * d * h + 4 * L * d) + 2 * d * d + 2 * d * 64


def esmc(d, H, NL):
    from esm_amd.esmc import ESMC
    m = ESMC(d, H, NL)
    m.load_state_dict(dev_weights(synth.esmc_shapes(d, NL)))
    m = m.cuda()
    if os.environ.get("ESMK_SMALL_TOKENS"):   # experiments: move the small-batch path's token limit (product: 1024)
        m.transformer.small_tokens = int(os.environ["ESMK_SMALL_TOKENS"])
    m._engines()
    return m


def c1():
    m = esmc(960, 15, 30)
    tok = synth.synth_tokens([256] * 8).reshape(-1).cuda()
    batch = DeviceBatch(pack_lengths([256] * 8), "cuda")
    dt = timed(lambda: m.forward_packed(tok, batch), 20, 5)
    n = 8 * 256
    return {"config": "C1 ESM C 300M-shaped, 8 x 256", "ms_per_forward": dt * 1e3, "residues_per_s": n / dt,
            "mfma_frac": n / dt * flop_per_token(960, 2560, 30, 256) / PEAK}


def c3(B=16, L=512):
    from esm_amd.esm3 import ESM3
    d, H, VH, NL = 1536, 24, 256, 48
    m = ESM3(d, H, VH, NL)
    m.load_state_dict(dev_weights(synth.esm3_shapes(d, NL, VH)))
    m = m.cuda()
    if os.environ.get("ESMK_SMALL_TOKENS"):
        m.transformer.small_tokens = int(os.environ["ESMK_SMALL_TOKENS"])
    seq = synth.synth_tokens([L] * B).cuda()
    g = torch.Generator().manual_seed(1)
    struct = torch.randint(0, 4096, (B, L), generator=g).cuda()
    coords = synth.synth_backbone(B, L, seed=3).cuda()
    fn = lambda: m(sequence_tokens=seq, structure_tokens=struct, structure_coords=coords)
    dt = timed(fn, 5 if B > 1 else 20, 2)
    n = B * L
    fl = flop_per_token(d, 4096, NL, L) + 2 * d * (15 * VH + 3 * VH) + 5 * 2 * d * d + 2 * d * (4096 + 11 + 19 + 2080 + 1478)
    return {"config": f"C3 ESM3-open-shaped (48 layers, geometric block 0), {B} x {L}, seq+struct+coords, all 6 heads",
            "ms_per_forward": dt * 1e3, "residues_per_s": n / dt, "mfma_frac": n / dt * fl / PEAK}


def c4(n_seq=512, budget=32768):
    m = esmc(1152, 18, 36)
    rng = np.random.default_rng(7)
    lengths = np.exp(rng.uniform(np.log(64), np.log(2048), n_seq)).astype(int)
    packs = make_packs(lengths, budget)
    batches = []
    for p in packs:
        ls = [int(lengths[i]) for i in p]
        tok = synth.synth_tokens(ls, seed=1).reshape(-1)
        keep = torch.cat([torch.arange(n) + j * max(ls) for j, n in enumerate(ls)])
        batches.append((tok[keep].cuda(), DeviceBatch(pack_lengths(ls), "cuda")))

    def fn():
        for tok, b in batches:
            m.forward_packed(tok, b, want_logits=False)
    dt = timed(fn, 3, 1)
    n = int(lengths.sum())
    fl = sum(flop_per_token(1152, 3072, 36, int(l)) * int(l) for l in lengths)
    return {"config": f"C4 ESM C 600M-shaped, {n_seq} sequences L in [64,2048] log-uniform, {len(packs)} packs <= {budget} tokens",
            "ms_total": dt * 1e3, "residues_per_s": n / dt, "mfma_frac": fl / dt / PEAK}


def c5(B=4, L=4096, NL=80):
    from esm_amd.transformer_stack import TransformerStack
    d, H = 2560, 40
    m = TransformerStack(d, H, None, NL, n_layers_geom=0)
    m.load_state_dict(dev_weights(synth.trunk_shapes("", d, NL)))
    m = m.cuda()
    eng = m.engine()
    batch = DeviceBatch(pack_lengths([L] * B), "cuda")
    x0 = torch.randn(B * L, d, device="cuda")

    def fn():
        eng.forward(x0.clone(), batch, want_post=True, want_hiddens=False)
    dt = timed(fn, 3, 1)
    n = B * L
    fl = NL * (8 * d * d + 6 * d * 6912 + 4 * L * d)
    return {"config": f"C5 6B-shaped trunk (d2560, 40 heads, {NL} layers), {B} x {L}", "ms_per_forward": dt * 1e3,
            "residues_per_s": n / dt, "mfma_frac": n / dt * fl / PEAK}


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c3", "c4", "c5"]
    for w in which:
        r = {"c1": c1, "c3": c3, "c3b1": lambda: c3(B=1), "c4": c4, "c5": c5}[w]()
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()
