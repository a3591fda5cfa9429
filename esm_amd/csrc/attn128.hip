// Attention for the function-token decoder's stack (reference
// esm/models/function_decoder.py:92-103,172-174 -> esm/layers/attention.py:48-82 with qk_layernorm=False,
// sequence_id=None): sequences of exactly 64 tokens (8 function tokens x 8 LSH bits), head dim 128, no mask,
// rotary over the full 128-wide head (esm/layers/rotary.py:36-65).  One workgroup per (sequence, head):
// Q, K (rotated while staging) and V live in 51 KB of LDS; wave w owns queries 16w..16w+15.
//   S^T = K . Q^T   MFMA 16x16x32 bf16, K rows fed in a permuted order so that the accumulators ARE the B operand
//   P   = softmax   (exp2; the softmax scale and log2 e are folded into Q)
//   O^T = V^T . P^T
#include "common.hpp"

namespace {

constexpr int AL = 64, AD = 128;
constexpr int AP = AD + 8;  // LDS row pitch in bf16 (272 B: 16 rows of an MFMA operand spread over the banks)

__global__ __launch_bounds__(256) void attn_l64_d128_kernel(const __bf16* __restrict__ qkv, int d,
                                                            const float* __restrict__ rope_cos,
                                                            const float* __restrict__ rope_sin, float q_scale,
                                                            __bf16* __restrict__ ctx) {
    __shared__ __attribute__((aligned(16))) __bf16 Qs[AL * AP];
    __shared__ __attribute__((aligned(16))) __bf16 Ks[AL * AP];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[AL * AP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * AL;
    const int h = blockIdx.y;
    const int ld = 3 * d;
    // ---- stage: rotary on q, k (pairs j, j + 64 of the head), plain copy of v ----
    for (int it = tid; it < AL * 8; it += 256) {
        const int r = it >> 3, j0 = (it & 7) * 8;  // row (= rotary position), first of 8 frequencies
        const float* c = rope_cos + r * (AD / 2) + j0;
        const float* s = rope_sin + r * (AD / 2) + j0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const __bf16* src = qkv + (row0 + r) * ld + which * d + h * AD;
            const bf16x8 x1 = *(const bf16x8*)(src + j0), x2 = *(const bf16x8*)(src + AD / 2 + j0);
            bf16x8 o1, o2;
            const float sc = which == 0 ? q_scale : 1.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the reference rotates in the tensor's dtype (bf16 tables would round; fp32 here) - rotary.py:57-65
                const float a = (float)x1[e], b = (float)x2[e];
                o1[e] = (__bf16)((a * c[e] - b * s[e]) * sc);
                o2[e] = (__bf16)((b * c[e] + a * s[e]) * sc);
            }
            __bf16* dst = (which == 0 ? Qs : Ks) + r * AP;
            *(bf16x8*)(dst + j0) = o1;
            *(bf16x8*)(dst + AD / 2 + j0) = o2;
        }
    }
    for (int it = tid; it < AL * 16; it += 256) {
        const int r = it >> 4, c8 = (it & 15) * 8;
        *(bf16x8*)(Vs + r * AP + c8) = *(const bf16x8*)(qkv + (row0 + r) * ld + 2 * d + h * AD + c8);
    }
    __syncthreads();
    // ---- S^T = K . Q^T : block m holds keys 32 (m >> 1) + 8 g + 4 (m & 1) + j of query 16 w + n ----
    f32x4 s4[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) s4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 qf = *(const bf16x8*)(Qs + (16 * w + n) * AP + 32 * kk + 8 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int key = 32 * (m >> 1) + 8 * (n >> 2) + 4 * (m & 1) + (n & 3);
            const bf16x8 kf = *(const bf16x8*)(Ks + key * AP + 32 * kk + 8 * g);
            s4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, s4[m], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s4[m][j]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s4[m][j] = __builtin_amdgcn_exp2f(s4[m][j] - mx);
            sum += s4[m][j];
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // ---- O^T = V^T . P^T ----
    f32x4 o4[8];
#pragma unroll
    for (int md = 0; md < 8; ++md) o4[md] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pf[j] = (__bf16)s4[2 * kk][j];
            pf[4 + j] = (__bf16)s4[2 * kk + 1][j];
        }
#pragma unroll
        for (int md = 0; md < 8; ++md) {
            bf16x8 vf;  // V^T fragment: dim 16 md + n, keys 32 kk + 8 g + 0..7
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = Vs[(32 * kk + 8 * g + e) * AP + 16 * md + n];
            o4[md] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o4[md], 0, 0, 0);
        }
    }
    const float inv = 1.0f / sum;
    __bf16* dst = ctx + (row0 + 16 * w + n) * d + h * AD + 4 * g;
#pragma unroll
    for (int md = 0; md < 8; ++md) {
        u32x2 pk;
        pk.x = pack_bf16x2(o4[md][0] * inv, o4[md][1] * inv);
        pk.y = pack_bf16x2(o4[md][2] * inv, o4[md][3] * inv);
        *(u32x2*)(dst + 16 * md) = pk;
    }
}

}  // namespace

extern "C" int esmk_attn_l64_d128(const void* qkv_bf16, void* ctx_bf16, const float* rope_cos, const float* rope_sin,
                                  int n_sequences, int d, int n_heads, esmk_stream_t stream) {
    if (n_sequences == 0) return ESMK_OK;
    if (!qkv_bf16 || !ctx_bf16 || !rope_cos || !rope_sin) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_l64_d128: null pointer");
    if (n_sequences < 0 || n_heads <= 0 || n_heads > 65535 || d != n_heads * AD)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_attn_l64_d128: need d == 128 * n_heads (d=%d, n_heads=%d)", d, n_heads);
    // softmax scale dh^-1/2 (torch SDPA default, attention.py:70-78) and log2(e) for exp2
    const float q_scale = 0.08838834764831845f * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_l64_d128_kernel, dim3(n_sequences, n_heads), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)qkv_bf16, d, rope_cos, rope_sin, q_scale, (__bf16*)ctx_bf16);
    ESMK_CHECK_LAUNCH("esmk_attn_l64_d128");
    return ESMK_OK;
}
