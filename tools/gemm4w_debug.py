"""Error map of the four-wave GEMM kernel on single tiles (run with ESMK_GEMM_TILE=256 ESMK_GEMM_4W=1)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esm_amd import _lib as L
from esm_amd import engine


def main():
    lib = L.lib_experiments()  # the ESMK_GEMM_* switches exist in libesmk_x.so only
    st = torch.cuda.current_stream().cuda_stream
    for M, N, K in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 256), (512, 512, 192), (200, 256, 192)]:
        g = torch.Generator().manual_seed(1)
        A = torch.randn(M, K, generator=g).bfloat16().cuda()
        W = torch.randn(N, K, generator=g) * 0.1
        Wp = engine.pack_weight(W, "cuda")
        ref = A.float() @ W.bfloat16().float().cuda().T
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_BF16, None, 1.0, out.data_ptr(), st))
        torch.cuda.synchronize()
        err = (out.float() - ref).abs()
        err = torch.nan_to_num(err, nan=99.0)
        bad = err > 0.02 * ref.abs().max() + 1e-3
        print(f"M={M} N={N} K={K}: max err {err.max().item():.4f}  bad {bad.float().mean().item():.4f}", flush=True)
        if bad.any():
            Mp, Np = (M + 15) // 16 * 16, (N + 15) // 16 * 16
            b = torch.zeros(Mp, Np, dtype=torch.bool, device="cuda")
            b[:M, :N] = bad
            blk = b.view(Mp // 16, 16, Np // 16, 16).any(3).any(1).cpu()
            print("  16x16 blocks with errors (rows = token blocks, cols = channel blocks):")
            for r in range(blk.shape[0]):
                print("  ", "".join("X" if x else "." for x in blk[r].tolist()))
            # out32 path too
        o32 = torch.full((M, N), float("nan"), device="cuda")
        L.check(lib.esmk_gemm(A.data_ptr(), K, Wp.data_ptr(), M, N, K, L.EPI_F32, None, 1.0, o32.data_ptr(), st))
        torch.cuda.synchronize()
        e32 = torch.nan_to_num((o32 - ref).abs(), nan=99.0)
        print(f"   f32 epilogue: max err {e32.max().item():.5f}", flush=True)


if __name__ == "__main__":
    main()
