"""End-to-end embedding throughput, strings in -> pooled embeddings out (ESM C 600M-shaped, random init):
tokenisation, pack planning, pinned uploads, trunk, mean pooling.  Compare with bench.py (trunk only).
   python tools/embed_bench.py [n_sequences] [fixed|mixed]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from esm_amd import synth
from esm_amd.esmc import ESMC
from esm_amd.pipeline import embed_sequences
from tools.bench_configs import dev_weights


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    mode = sys.argv[2] if len(sys.argv) > 2 else "fixed"
    rng = np.random.default_rng(0)
    lens = np.full(n, 510) if mode == "fixed" else np.exp(rng.uniform(np.log(62), np.log(2046), n)).astype(int)
    aa = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    seqs = ["".join(aa[rng.integers(0, 20, size=int(k))]) for k in lens]
    m = ESMC(1152, 18, 36)
    m.load_state_dict(dev_weights(synth.esmc_shapes(1152, 36)))
    m = m.cuda()
    embed_sequences(m, seqs[:128])  # warm-up (packs weights, sizes workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = embed_sequences(m, seqs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    m.tokenizer.encode_packed(seqs)
    t_tok = time.perf_counter() - t1
    res = int(lens.sum()) + 2 * n
    print(json.dumps({"sequences": n, "lengths": mode, "residues": res, "seconds": dt, "residues_per_s": res / dt,
                      "sequences_per_s": n / dt, "tokenize_seconds": t_tok, "out_shape": list(out.shape)}))


if __name__ == "__main__":
    main()
