"""Does a HIP graph of the trunk forward help small batches?  One esmk_trunk_forward call (48 blocks, d 1536, 1 x 512 and
8 x 256 d 960 x 30) captured with torch.cuda.CUDAGraph and replayed, against the eager call.   python tools/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import synth
from esm_amd.engine import DeviceBatch
from esm_amd.packing import pack_lengths
from esm_amd.transformer_stack import TransformerStack

def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for d, H, NL, lens in ((1536, 24, 48, [512]), (960, 15, 30, [256] * 8), (1152, 18, 36, [512] * 64)):
    g = torch.Generator(device="cuda").manual_seed(0)
    wd = {}
    for k, shp in synth.trunk_shapes("", d, NL).items():
        mean, std = synth._kind(k)
        wd[k] = torch.randn(shp, device="cuda", generator=g) * std + mean
    m = TransformerStack(d, H, None, NL, n_layers_geom=0)
    m.load_state_dict(wd)
    del wd
    eng = m.cuda().engine()
    batch = DeviceBatch(pack_lengths(lens), "cuda")
    N = sum(lens)
    x0 = torch.randn(N, d, device="cuda")
    x = x0.clone()
    def eager():
        x.copy_(x0)
        return eng.forward(x, batch, want_post=True, want_hiddens=False)
    post_e = eager()[0].clone()
    t_e = timed(eager)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eager()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            out = eager()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    same = torch.equal(out[0], post_e)
    t_g = timed(gr.replay)
    print(f"d={d} layers={NL} tokens={N}: eager {t_e:.3f} ms  graph replay {t_g:.3f} ms  ({(t_e / t_g - 1) * 100:+.1f} %)  identical={same}", flush=True)
    del eng, m
    torch.cuda.empty_cache()
