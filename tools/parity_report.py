"""Parity of the HIP path on the full-depth fixtures next to the reference's own bf16-vs-fp32 drift on the same
inputs (stored in the fixtures by oracle/make_golden.py).   python tools/parity_report.py"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _golden as G
from esm_amd.esmc import ESMC

for name in ("esmc_300m_2x128", "esmc_600m_2x128"):
    fx = G.load(name)
    m = ESMC(int(fx["d"]), int(fx["H"]), int(fx["NL"]))
    m.load_state_dict(G.weights_for(fx))
    m = m.cuda()
    tok = G.t(fx["tokens"])
    out = m(tok.cuda())
    s = int(fx["token_stride"])
    valid = (tok != 1)[:, ::s]
    emb, ref_emb = out.embeddings.cpu()[:, ::s][valid], G.t(fx["embeddings"])[valid]
    lg, ref_lg = out.sequence_logits.cpu()[:, ::s][valid], G.t(fx["sequence_logits"])[valid]
    cos = torch.nn.functional.cosine_similarity(emb, ref_emb, dim=-1).min().item()
    dl = (lg - ref_lg).abs().max().item()
    gate = 0.05 + 0.02 * ref_lg.abs().max().item()
    print(json.dumps({"fixture": name, "ours_cos_min": cos, "ours_logit_maxdiff": dl, "logit_gate": gate,
                      "reference_bf16_cos_min": float(fx["ref_bf16_cos_min"]),
                      "reference_bf16_logit_maxdiff": float(fx["ref_bf16_logit_maxdiff"])}))
