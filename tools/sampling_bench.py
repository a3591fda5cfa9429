"""One sampling step over a batch of rows: fused esmk_sample_track vs the reference's chain of torch ops
(sort/softmax/cumsum top-p, masks, softmax, multinomial, log_softmax, entropy) run on the same GPU.
   python tools/sampling_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from esm_amd.sampling import SamplingTrackConfig, get_sampling_mask, sample_track


def torch_chain(logits, tokens, cfg, mask_idx, valid_ids):
    """The reference formulation (sampling.py:153-200,293-313; generation.py:703-733,786-822) in plain torch."""
    V = logits.shape[-1]
    flat = logits.reshape(-1, V)
    if cfg.top_p < 1.0:
        sl, si = torch.sort(flat, dim=-1, descending=True)
        keep = sl.softmax(-1).cumsum(-1) <= cfg.top_p
        keep[:, 0] = True
        b, _ = torch.where(~keep)
        flat[b, si[~keep]] = torch.finfo(flat.dtype).min
    m = torch.ones_like(flat, dtype=torch.bool)
    m[..., valid_ids] = False
    flat[m] = -torch.inf
    probs = torch.softmax(flat / cfg.temperature, -1)
    ids = torch.multinomial(probs, 1).squeeze(1).reshape(tokens.shape)
    lp = logits.log_softmax(-1)
    sm = get_sampling_mask(tokens, cfg, mask_idx)
    ids = torch.where(sm, ids, tokens)
    ent = torch.distributions.Categorical(logits=lp).entropy()
    slp = torch.zeros_like(ids, dtype=lp.dtype)
    slp[sm] = lp[sm, ids[sm]]
    return ids, ent, slp, lp.exp().max(-1).values


def main():
    for V, mask_idx, B, Ln in ((64, 32, 16, 512), (4101, 4096, 16, 512), (4101, 4096, 1, 512)):
        logits0 = torch.randn(B, Ln, V, device="cuda") * 3
        tokens = torch.full((B, Ln), mask_idx, device="cuda")
        tokens[:, 0], tokens[:, -1] = 0, 2
        # every id valid: with top-p AND invalid ids AND T < 1 the reference formulation can produce an all -inf row
        # (finfo.min / T overflows) and torch.multinomial then asserts on the device; the fused kernel does not
        valid = list(range(V))
        cfg = SamplingTrackConfig(temperature=0.7, top_p=0.9)
        res = {}
        for name, fn in (("fused", lambda l: sample_track(l, tokens, cfg, mask_idx, valid)),
                         ("torch_ops", lambda l: torch_chain(l, tokens, cfg, mask_idx, valid))):
            for _ in range(2):
                fn(logits0.clone())
            torch.cuda.synchronize()
            n = 10
            ls = [logits0.clone() for _ in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for l in ls:
                fn(l)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e3
        print(f"V={V} rows={B * Ln}: fused {res['fused']:.3f} ms   torch op chain {res['torch_ops']:.3f} ms   x{res['torch_ops'] / res['fused']:.1f}", flush=True)


if __name__ == "__main__":
    main()
