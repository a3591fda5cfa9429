"""esmk_k_prep: product library against variant builds (make VARIANT=name EXTRA=-D...) in one process; k comes from
HBM (256 MB written between calls).   python tools/kprep_ab.py kpNOROT kpNOSTORE ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine
libs = [("product", L.lib())] + [(n, L._load(os.path.join(os.path.dirname(L.LIB_PATH), f"libesmk_x_{n}.so"))) for n in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
M, H = 32768, 18
d = 64 * H
qkv = torch.randn(M, 3 * d, device="cuda").bfloat16()
gk = torch.ones(d, device="cuda")
pos = (torch.arange(M, device="cuda") % 512).int()
cos, sin = [t.cuda() for t in engine.rotary_tables(512)]
stats = torch.randn(M, 8, 2, device="cuda").abs()
qmr = torch.empty(M, 2, device="cuda")
other = torch.empty(64 << 20, device="cuda")
def run(lib, with_stats=True):
    L.check(lib.esmk_k_prep(qkv.data_ptr(), gk.data_ptr(), None, pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), 512, M, d, H, 1e-5,
                            stats.data_ptr() if with_stats else None, 8, qmr.data_ptr(), st))
res = {}
for r in range(7):
    for n, lib in libs + [("product-nostats", libs[0][1])]:
        qkv.normal_()
        other.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib, n != "product-nostats"); e1.record()
        torch.cuda.synchronize()
        res.setdefault(n, []).append(e0.elapsed_time(e1) * 1e3)
for n, t in res.items():
    print(f"{n:18s} median {sorted(t)[len(t)//2]:6.1f} us  min {min(t):6.1f}")
