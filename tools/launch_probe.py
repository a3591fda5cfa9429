"""Host enqueue time vs GPU time of one trunk forward (small batches: is the forward launch-bound?).
   python tools/launch_probe.py [B L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
from esm_amd import synth
from esm_amd.engine import DeviceBatch
from esm_amd.packing import pack_lengths
from esm_amd.transformer_stack import TransformerStack

B, Ln = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 512)
for name, d, H, NL in (("esm3-open trunk (no geometric block)", 1536, 24, 48), ("esmc-300m trunk", 960, 15, 30)):
    for small in (1 << 20, 0):  # small-batch path forced / off
        ts = TransformerStack(d, H, None, NL, n_layers_geom=0)
        ts.load_state_dict(bc.dev_weights({k[len("transformer."):]: v for k, v in synth.esmc_shapes(d, NL).items() if k.startswith("transformer.")}))
        ts.small_tokens = small
        ts = ts.cuda()
        eng = ts.engine()
        batch = DeviceBatch(pack_lengths([Ln] * B), "cuda")
        x0 = torch.randn(B * Ln, d, device="cuda")
        for _ in range(3):
            eng.forward(x0.clone(), batch)
        torch.cuda.synchronize()
        host, gpu = [], []
        for _ in range(10):
            x = x0.clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            eng.forward(x, batch)
            e1.record()
            host.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            gpu.append(e0.elapsed_time(e1))
        print(f"{name} {B}x{Ln} small_tokens={small}: host enqueue {sorted(host)[5]:.2f} ms, GPU {sorted(gpu)[5]:.2f} ms", flush=True)
