"""Iterative decoding around the trunk: ESM3-open-shaped model (random init), one fully masked prompt of length L,
`num_steps` decoding steps of the sequence track.  Reports time per step (forward + fused sampling + commit).
   python tools/generate_bench.py [L] [steps] [batch]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from esm_amd import synth
from esm_amd.api import ESMProteinTensor
from esm_amd.esm3 import ESM3
from esm_amd.sampling import GenerationConfig
from tools.bench_configs import dev_weights


def main():
    Ln = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    d, H, VH, NL = 1536, 24, 256, 48
    m = ESM3(d, H, VH, NL)
    m.load_state_dict(dev_weights(synth.esm3_shapes(d, NL, VH)))
    m = m.cuda()
    seq = torch.full((Ln,), 32)
    seq[0], seq[-1] = 0, 2
    prompts = [ESMProteinTensor(sequence=seq.clone()) for _ in range(B)]
    cfg = GenerationConfig(track="sequence", num_steps=steps, temperature=0.7)
    g = torch.Generator(device="cuda").manual_seed(0)
    m.generate_tokens(prompts, GenerationConfig(track="sequence", num_steps=2), generator=g)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate_tokens(prompts, cfg, generator=g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not any((o.sequence == 32).any() for o in out)
    print(json.dumps({"config": f"ESM3-open-shaped generate, sequence track, {B} x L={Ln}, {steps} steps",
                      "seconds": dt, "ms_per_step": dt / steps * 1e3, "residues_decoded_per_s": B * (Ln - 2) / dt}))


if __name__ == "__main__":
    main()
