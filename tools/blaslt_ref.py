"""What the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the trunk's GEMM shapes, as an outside
reference for the hand-written kernels (not used by the product path)."""
import torch

def main():
    for name, M, N, K in (("ffn1", 32768, 6144, 1152), ("qkv", 32768, 3456, 1152), ("ffn2", 32768, 1152, 3072), ("out", 32768, 1152, 1152),
                          ("big", 16384, 8192, 8192)):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        for _ in range(5):
            torch.matmul(a, w.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, w.t())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:5s} M={M} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s (torch.matmul bf16, plain output)", flush=True)
main()
