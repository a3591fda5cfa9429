"""k_prep / qk_prep times with the data coming from HBM (256 MB written between calls), plus a check against a copy
processed by torch in fp32.   python tools/rowpass_bench.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for M, H in ((32768, 18), (8192, 24), (16384, 40)):
    d = 64 * H
    qkv = torch.randn(M, 3 * d, device="cuda").bfloat16()
    gq, gk = torch.ones(d, device="cuda"), torch.ones(d, device="cuda")
    pos = (torch.arange(M, device="cuda") % 512).int()
    cos, sin = [t.cuda() for t in engine.rotary_tables(512)]
    stats = torch.randn(M, 8, 2, device="cuda").abs()
    qmr = torch.empty(M, 2, device="cuda")
    other = torch.empty(64 << 20, device="cuda")
    fk = lambda: L.check(lib.esmk_k_prep(qkv.data_ptr(), gk.data_ptr(), None, pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), 512, M, d, H,
                                         1e-5, stats.data_ptr(), 8, qmr.data_ptr(), st))
    fq = lambda: L.check(lib.esmk_qk_prep(qkv.data_ptr(), gq.data_ptr(), gk.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(), 512,
                                          M, d, H, 0.18, 1e-5, st))
    for name, f in (("k_prep", fk), ("qk_prep", fq)):
        ts = []
        for _ in range(9):
            qkv.normal_()
            other.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"{name:8s} M={M} d={d}: median {sorted(ts)[4]:.1f} us  min {min(ts):.1f} us", flush=True)
