"""Micro-benchmark of the HBM-bound row kernels at the C2 shape (32768 x 1152), rotating through several
buffer sets so that nothing is served from the 256 MB Infinity Cache.   python tools/rowops_bench.py [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = L.lib()
    M, d, H = 32768, 1152, 18
    st = torch.cuda.current_stream().cuda_stream
    nset = 6
    xs = [torch.randn(M, d, device="cuda") for _ in range(nset)]
    ds = [torch.randn(M, d, device="cuda").bfloat16() for _ in range(nset)]
    ys = [torch.empty(M, d, device="cuda", dtype=torch.bfloat16) for _ in range(nset)]
    qkvs = [torch.randn(M, 3 * d, device="cuda").bfloat16() for _ in range(nset)]
    g = torch.ones(d, device="cuda")
    b = torch.zeros(d, device="cuda")
    cos, sin = engine.rotary_tables(4096)
    cos, sin = cos.cuda(), sin.cuda()
    pos = (torch.arange(M, device="cuda", dtype=torch.int32) % 512).contiguous()

    def add_ln(i):
        L.check(lib.esmk_add_layernorm(xs[i].data_ptr(), ds[i].data_ptr(), g.data_ptr(), b.data_ptr(), ys[i].data_ptr(),
                                       L.BF16, M, d, 1e-5, st))

    def ln(i):
        L.check(lib.esmk_layernorm(xs[i].data_ptr(), L.F32, g.data_ptr(), b.data_ptr(), ys[i].data_ptr(), L.BF16, M, d, 1e-5, st))

    def qk(i):
        L.check(lib.esmk_qk_prep(qkvs[i].data_ptr(), g.data_ptr(), g.data_ptr(), pos.data_ptr(), cos.data_ptr(),
                                 sin.data_ptr(), 4096, M, d, H, 0.18, 1e-5, st))

    for name, fn, nbytes in (("add_layernorm", add_ln, M * d * (4 + 2 + 4 + 2)), ("layernorm", ln, M * d * (4 + 2)),
                             ("qk_prep", qk, M * d * 2 * 4)):
        for i in range(nset):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(iters):
            fn(k % nset)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name:14s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
