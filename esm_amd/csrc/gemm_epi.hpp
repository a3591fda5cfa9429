// Shared GEMM epilogues.  A lane holds 4 consecutive output channels (c..c+3) of ONE token t in an f32x4.
#pragma once
#include "common.hpp"

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// silu(x) = x * sigmoid(x) on the hardware exp2 / rcp units (about 1 ulp each; the result is rounded to bf16)
__device__ __forceinline__ float silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// SwiGLU: g/u = gate/up accumulators of the same 4 channels; c = OUTPUT channel, hw = output width.
// bias (optional) is ffn.1.bias in the reference's row order: gate rows [0, hw), up rows [hw, 2 hw).
__device__ __forceinline__ void epi_swiglu4(f32x4 g, f32x4 u, int t, int c, int hw, void* out,
                                            const float* __restrict__ bias = nullptr) {
    if (c >= hw) return;
    if (bias) {
        g += *(const f32x4*)(bias + c);  // hw % 16 == 0 (SwiGLU packing) and c % 4 == 0
        u += *(const f32x4*)(bias + hw + c);
    }
    u32x2 v;
    v.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
    v.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
    *(u32x2*)((__bf16*)out + (size_t)t * hw + c) = v;
}

template <int EPI>
__device__ __forceinline__ void epi_store4(f32x4 v, int t, int c, int N, const float* __restrict__ bias, float alpha,
                                           void* out) {
    if (c >= N) return;
    const bool full = ((N & 3) == 0);  // then c+3 < N as well
    if constexpr (EPI == ESMK_EPI_RESIDUAL) {
        float* o = (float*)out + (size_t)t * N + c;
        f32x4 x = *(f32x4*)o;
        if (bias) v += *(const f32x4*)(bias + c);  // N % 4 == 0 for this epilogue
        x += v * alpha;
        *(f32x4*)o = x;
    } else if constexpr (EPI == ESMK_EPI_BF16 || EPI == ESMK_EPI_BIAS_GELU) {
        __bf16* o = (__bf16*)out + (size_t)t * N + c;
        if constexpr (EPI == ESMK_EPI_BIAS_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r] + (c + r < N ? bias[c + r] : 0.f));
        } else {
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (c + r < N ? bias[c + r] : 0.f);
            }
            v = scale4(v, alpha);
        }
        if (full) {
            u32x2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *(u32x2*)o = pk;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c + r < N) o[r] = (__bf16)v[r];
        }
    } else {  // ESMK_EPI_F32
        float* o = (float*)out + (size_t)t * N + c;
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (c + r < N ? bias[c + r] : 0.f);
        }
        if (full) {
            *(f32x4*)o = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c + r < N) o[r] = v[r];
        }
    }
}
