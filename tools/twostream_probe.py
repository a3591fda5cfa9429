"""C2 forward as ONE stream over 64 sequences vs TWO streams over 32 sequences each (independent half batches:
the HBM-bound row kernels of one half run under the MFMA-bound GEMMs of the other).
   python tools/twostream_probe.py [n_splits ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import synth
from esm_amd.engine import DeviceBatch
from esm_amd.esmc import ESMC
from esm_amd.packing import pack_lengths

D, H, NL, B, SL = 1152, 18, 36, 64, 512


def main():
    splits = [int(a) for a in sys.argv[1:]] or [1, 2]
    dev = torch.device("cuda", 0)
    model = ESMC(D, H, NL)
    model.load_state_dict(synth.esmc_weights(D, NL, seed=0))
    model = model.to(dev)
    trunk, head = model._engines()
    tokens = synth.synth_tokens([SL] * B, seed=1234).reshape(-1).to(dev)
    for ns in splits:
        nb = B // ns
        parts = [tokens[i * nb * SL:(i + 1) * nb * SL].contiguous() for i in range(ns)]
        batches = [DeviceBatch(pack_lengths([SL] * nb), dev) for _ in range(ns)]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        ws_t, ws_h = [None] * ns, [None] * ns

        def step():
            cur = torch.cuda.current_stream()
            for i in range(ns):
                streams[i].wait_stream(cur)
                trunk._ws, head._ws = ws_t[i], ws_h[i]
                with torch.cuda.stream(streams[i]):
                    model.forward_packed(parts[i], batches[i], want_hiddens=False, want_logits=True)
                ws_t[i], ws_h[i] = trunk._ws, head._ws
            for i in range(ns):
                cur.wait_stream(streams[i])

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"splits={ns}: {ms:.2f} ms/step  {B * SL / ms * 1e3:.0f} residues/s", flush=True)


if __name__ == "__main__":
    main()
