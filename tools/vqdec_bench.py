"""Structure decoder timing on one GPU (released geometry: d 1280, 20 heads, 30 blocks; random weights).

Prints one JSON line per (B, L): whole decode() and the fused pair kernel alone (HIP events on the launch stream =
torch's current stream), next to the reference's formulation of the pair head + pae + ptm written as the torch op
chain it uses (vqvae.py:79-95, predicted_aligned_error.py:37-66) on the same device in fp32 and bf16.
Usage: python tools/vqdec_bench.py [--cases 1x512,16x512,4x1024]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import _lib as L  # noqa: E402
from esm_amd import synth  # noqa: E402
from esm_amd.engine import _stream  # noqa: E402
from esm_amd.vqvae import StructureTokenDecoder  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def torch_pair_chain(x, w, aa, dtype):
    """The reference's op chain for the pair head, pae and ptm, on the GPU."""
    F = torch.nn.functional
    p = "pairwise_classification_head."
    x = x.to(dtype)
    z = x @ w[p + "downproject.weight"].to(dtype).T
    q, k = z.chunk(2, dim=-1)
    feat = torch.cat([q[:, None, :, :] * k[:, :, None, :], q[:, None, :, :] - k[:, :, None, :]], dim=-1)
    h = F.gelu(feat @ w[p + "linear1.weight"].to(dtype).T)
    h = F.layer_norm(h, (128,), w[p + "norm.weight"].to(dtype), w[p + "norm.bias"].to(dtype))
    logits = (h @ w[p + "linear2.weight"].to(dtype).T).split([64, 96, 64], dim=-1)[2]
    step = 31.0 / 62
    bins = (torch.arange(64, device=x.device, dtype=torch.float32) + 0.5) * step
    sq = aa[:, :, None] & aa[:, None, :]
    probs = logits.masked_fill(~sq[..., None], torch.finfo(logits.dtype).min).softmax(-1)
    pae = (probs * bins).sum(-1)
    d0 = 1.24 * (aa.sum(-1, keepdim=True).clamp_min(19) - 15) ** (1 / 3) - 1.8
    f_d = 1.0 / (1 + (bins / d0.unsqueeze(-1)) ** 2)
    tm = (probs * f_d.unsqueeze(-2)).sum(-1)
    tm = (sq * tm).sum(-1) / (1e-10 + sq.sum(-1))
    return pae, tm.max(-1).values


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="1x512,16x512,4x1024")
    args = ap.parse_args()
    d, H, NL = 1280, 20, 30
    w = synth.make_weights(synth.vqdec_shapes(d, NL), seed=5)
    m = StructureTokenDecoder(d, H, NL)
    m.load_state_dict(w, strict=True)
    m = m.cuda()
    wd = {k: v.cuda() for k, v in w.items() if k.startswith("pairwise_classification_head.")}
    lib = L.lib()
    for case in args.cases.split(","):
        B, Ln = (int(v) for v in case.split("x"))
        g = torch.Generator().manual_seed(B * 1000 + Ln)
        tok = torch.randint(0, 4096, (B, Ln), generator=g)
        tok[:, 0], tok[:, -1] = 4098, 4097
        tok = tok.cuda()
        ms_decode = timed(lambda: m.decode(tok))
        _, e = m._engines()
        qk = torch.randn(B * Ln, 128, device="cuda")
        aa = (tok < 4096).to(torch.uint8).contiguous()
        pae = torch.empty(B, Ln, Ln, device="cuda")
        ptm = torch.empty(B, device="cuda")
        ms_pair = timed(lambda: L.check(lib.esmk_pairwise_pae(L.ptr(qk), L.ptr(e["w1"]), L.ptr(e["w2"]), L.ptr(e["b2"]),
                                                             L.ptr(aa), B, Ln, 31.0, L.ptr(pae), L.ptr(ptm), _stream())), n=20)
        x = torch.randn(B, Ln, d, device="cuda")
        rec = {"B": B, "L": Ln, "decode_ms": round(ms_decode, 3), "residues_per_s": round(B * Ln / ms_decode * 1e3),
               "pair_kernel_ms": round(ms_pair, 4),
               "pair_kernel_tflops": round(B * Ln * Ln * (2 * 128 * 128 + 2 * 128 * 64) / ms_pair / 1e9, 1)}
        for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            try:
                rec[f"torch_chain_{name}_ms"] = round(timed(lambda: torch_pair_chain(x, wd, aa.bool(), dt), n=5, warm=2), 3)
            except torch.OutOfMemoryError:
                rec[f"torch_chain_{name}_ms"] = None
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
