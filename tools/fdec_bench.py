"""Function-token decoder timing on one GPU (released geometry, random weights) + error against the d1024 fixture."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd import synth  # noqa: E402
from esm_amd.function_decoder import FunctionTokenDecoder  # noqa: E402
from tests import _golden as G  # noqa: E402


def main():
    fx = G.load("fdec_d1024_3l")
    m = FunctionTokenDecoder()
    m.load_state_dict(G.weights_for(fx), strict=True)
    m = m.cuda()
    out = m(G.t(fx["tokens"]).cuda())
    errs = {k: float(np.abs(out[k].cpu().numpy()[:, ::int(fx["stride"])] - fx[k]).max()) for k in ("keyword_logits", "keyword_tfidf", "interpro_logits")}
    print(json.dumps({"max_err_vs_fp32_reference": errs, "reference_bf16_drift": {k: float(fx["ref_bf16_maxdiff_" + k]) for k in errs}}))
    for B in (512, 2048):
        tok = torch.randint(4, 260, (B, 8)).cuda()
        for _ in range(3):
            m(tok)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            m(tok)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(json.dumps({"residues": B, "forward_ms": round(ms, 3), "residues_per_s": round(B / ms * 1e3)}), flush=True)


if __name__ == "__main__":
    with torch.no_grad():
        main()
