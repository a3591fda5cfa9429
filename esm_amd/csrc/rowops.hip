// Row-wise, HBM-bound kernels: LayerNorm, QK-LayerNorm + rotary, embedding gathers.
#include "common.hpp"

namespace {

// --------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (two-pass mean / variance in fp32).
// --------------------------------------------------------------------------------------------------
constexpr int LN_MAX_VEC = 16;  // 16 float4 per lane -> d <= 4096

// ADD: x (f32, in/out) += delta (bf16) before normalising; the updated row is written back to x.
template <typename TIN, typename TOUT, bool ADD = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TOUT* __restrict__ y, int M,
                                                        int d, int ldy, float eps,
                                                        const __bf16* __restrict__ delta = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = d >> 2;  // d % 4 == 0
    f32x4 v[LN_MAX_VEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            if constexpr (sizeof(TIN) == 4) {
                v[i] = *(const f32x4*)((const float*)x + (size_t)row * d + c * 4);
            } else {
                const u32x2 p = *(const u32x2*)((const __bf16*)x + (size_t)row * d + c * 4);
                v[i] = f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u),
                             __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u)};
            }
            if constexpr (ADD) {
                const u32x2 p = *(const u32x2*)(delta + (size_t)row * d + c * 4);
                v[i] += f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u),
                              __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u)};
                *(f32x4*)((float*)x + (size_t)row * d + c * 4) = v[i];
            }
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] -= mean;
            sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < LN_MAX_VEC; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const f32x4 g = *(const f32x4*)(gamma + c * 4);
            f32x4 o = v[i] * rstd * g;
            if (beta) o += *(const f32x4*)(beta + c * 4);
            if constexpr (sizeof(TOUT) == 4) {
                *(f32x4*)((float*)y + (size_t)row * ldy + c * 4) = o;
            } else {
                u32x2 p;
                p.x = pack_bf16x2(o[0], o[1]);
                p.y = pack_bf16x2(o[2], o[3]);
                *(u32x2*)((__bf16*)y + (size_t)row * ldy + c * 4) = p;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------
// QK-LayerNorm (over the full d_model width, bias-free) + NeoX rotary per 64-wide head, in place.
// One WAVE per token (4 tokens per 256-thread block): the q and k rows live in registers, statistics are
// wave shuffles - no LDS, no block barrier.  A "unit" is (q|k, head, 8-column chunk c in 0..3) and covers
// columns [8c, 8c+8) and their rotary partners [32+8c, 32+8c+8) of that head: 8*H units per token.
// --------------------------------------------------------------------------------------------------
constexpr int QK_MAX_UNITS = 6;  // per lane -> 8*H <= 384 -> H <= 48

__device__ __forceinline__ void unpack8(const u32x4 p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(p[e] << 16);
        f[2 * e + 1] = __uint_as_float(p[e] & 0xffff0000u);
    }
}

template <int NU>  // units per lane (compile-time so the row stays in registers)
__global__ __launch_bounds__(256) void qk_prep_kernel(__bf16* __restrict__ qkv, const float* __restrict__ gq,
                                                      const float* __restrict__ gk,
                                                      const int* __restrict__ positions,
                                                      const float* __restrict__ rcos,
                                                      const float* __restrict__ rsin, int rope_len, int M, int d,
                                                      int H, float q_scale, float eps) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= M) return;
    __bf16* row = qkv + (size_t)tok * 3 * d;
    const int units_half = 4 * H;  // units for q; same count for k
    const int nunits = 2 * units_half;
    float lo[NU][8], hi[NU][8];
    float sq_ = 0.f, sk_ = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int unit = lane + u * 64;
        if (unit < nunits) {
            const int isk = unit >= units_half;
            const int uu = unit - isk * units_half;
            const int col = isk * d + (uu >> 2) * 64 + (uu & 3) * 8;
            unpack8(*(const u32x4*)(row + col), lo[u]);
            unpack8(*(const u32x4*)(row + col + 32), hi[u]);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += lo[u][e] + hi[u][e];
            if (isk) sk_ += s; else sq_ += s;
        }
    }
    const float inv_d = 1.0f / (float)d;
    const float mean_q = wave_sum(sq_) * inv_d;
    const float mean_k = wave_sum(sk_) * inv_d;
    sq_ = 0.f; sk_ = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int unit = lane + u * 64;
        if (unit < nunits) {
            const int isk = unit >= units_half;
            const float m = isk ? mean_k : mean_q;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lo[u][e] -= m;
                hi[u][e] -= m;
                s += lo[u][e] * lo[u][e] + hi[u][e] * hi[u][e];
            }
            if (isk) sk_ += s; else sq_ += s;
        }
    }
    const float rstd_q = rsqrtf(wave_sum(sq_) * inv_d + eps);
    const float rstd_k = rsqrtf(wave_sum(sk_) * inv_d + eps);
    int pos = positions[tok];
    pos = pos < rope_len ? pos : rope_len - 1;  // host validates; clamp keeps the load in bounds
    const float* cs = rcos + (size_t)pos * 32;
    const float* sn = rsin + (size_t)pos * 32;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int unit = lane + u * 64;
        if (unit < nunits) {
            const int isk = unit >= units_half;
            const int uu = unit - isk * units_half;
            const int hc = (uu >> 2) * 64 + (uu & 3) * 8;  // column inside q (or k)
            const int j0 = (uu & 3) * 8;                   // rotary frequency index of element 0
            const float* g = (isk ? gk : gq) + hc;
            const float rs = isk ? rstd_k : rstd_q;
            const float osc = isk ? 1.0f : q_scale;
            const f32x4 c0 = *(const f32x4*)(cs + j0), c1 = *(const f32x4*)(cs + j0 + 4);
            const f32x4 s0 = *(const f32x4*)(sn + j0), s1 = *(const f32x4*)(sn + j0 + 4);
            const f32x4 ga0 = *(const f32x4*)g, ga1 = *(const f32x4*)(g + 4);
            const f32x4 gb0 = *(const f32x4*)(g + 32), gb1 = *(const f32x4*)(g + 36);
            float o1[8], o2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = lo[u][e] * rs * (e < 4 ? ga0[e & 3] : ga1[e & 3]);
                const float x2 = hi[u][e] * rs * (e < 4 ? gb0[e & 3] : gb1[e & 3]);
                const float c = e < 4 ? c0[e & 3] : c1[e & 3], sv = e < 4 ? s0[e & 3] : s1[e & 3];
                o1[e] = (x1 * c - x2 * sv) * osc;
                o2[e] = (x2 * c + x1 * sv) * osc;
            }
            const int col = isk * d + hc;
            *(u32x4*)(row + col) = u32x4{pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3]),
                                         pack_bf16x2(o1[4], o1[5]), pack_bf16x2(o1[6], o1[7])};
            *(u32x4*)(row + col + 32) = u32x4{pack_bf16x2(o2[0], o2[1]), pack_bf16x2(o2[2], o2[3]),
                                              pack_bf16x2(o2[4], o2[5]), pack_bf16x2(o2[6], o2[7])};
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Embedding gather (ESM C) and the 8-track ESM3 input encoder.
// --------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table, int vocab,
                             float* __restrict__ out, int M, int d) {
    const int nvec = d >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * nvec) return;
    const int t = (int)(idx / nvec), c = (int)(idx - (size_t)t * nvec);
    long long tok = tokens[t];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);  // host validates the range; keep loads in bounds
    *(f32x4*)(out + (size_t)t * d + c * 4) = *(const f32x4*)(table + (size_t)tok * d + c * 4);
}

__device__ __forceinline__ long long clampi(long long v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

__global__ void esm3_encode_kernel(esmk_esm3_encoder enc, const int64_t* __restrict__ seq,
                                   const int64_t* __restrict__ st, const float* __restrict__ avg_plddt,
                                   const float* __restrict__ res_plddt, const int64_t* __restrict__ ss8,
                                   const int64_t* __restrict__ sasa, const int64_t* __restrict__ func,
                                   const int64_t* __restrict__ resid, float* __restrict__ out, int M, int d) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * d) return;
    const int t = (int)(idx / d), c = (int)(idx - (size_t)t * d);
    // summation order follows the reference expression (esm3.py:137-146)
    float acc = enc.sequence_embed[clampi(seq[t], 64) * d + c];
    {
        const float v = avg_plddt[t];
        float s = enc.plddt_b[c];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const float z = (v - (float)b * (1.0f / 15.0f)) * 16.0f;
            s += expf(-z * z) * enc.plddt_w[c * 16 + b];
        }
        acc += s;
    }
    {
        const float v = res_plddt[t];
        float s = enc.res_plddt_b[c];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const float z = (v - (float)b * (1.0f / 15.0f)) * 16.0f;
            s += expf(-z * z) * enc.res_plddt_w[c * 16 + b];
        }
        acc += s;
    }
    acc += enc.structure_embed[clampi(st[t], 4101) * d + c];
    acc += enc.ss8_embed[clampi(ss8[t], 11) * d + c];
    acc += enc.sasa_embed[clampi(sasa[t], 19) * d + c];
    {
        const int d8 = d >> 3;
        const int track = c / d8, cc = c - track * d8;
        const long long f = clampi(func[(size_t)t * 8 + track], 260);
        if (f != 0) acc += enc.function_embed[((size_t)track * 260 + f) * d8 + cc];
    }
    {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const long long a = clampi(resid[(size_t)t * 16 + r], 1478);
            if (a != 0) s += enc.residue_embed[a * d + c];
        }
        acc += s;
    }
    out[idx] = acc;
}

}  // namespace

extern "C" int esmk_layernorm(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                              int M, int d, float eps, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!x || !gamma || !y) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_layernorm: null pointer");
    if (M < 0 || d <= 0 || (d & 3) || d > LN_MAX_VEC * 256)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_layernorm: d=%d must be a multiple of 4 and <= %d", d, LN_MAX_VEC * 256);
    if (M == 0) return ESMK_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    if (x_dtype == ESMK_F32 && y_dtype == ESMK_BF16)
        hipLaunchKernelGGL((layernorm_kernel<float, __bf16>), grid, block, 0, s, (const float*)x, gamma, beta, (__bf16*)y, M, d, d, eps);
    else if (x_dtype == ESMK_F32 && y_dtype == ESMK_F32)
        hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, s, (const float*)x, gamma, beta, (float*)y, M, d, d, eps);
    else if (x_dtype == ESMK_BF16 && y_dtype == ESMK_BF16)
        hipLaunchKernelGGL((layernorm_kernel<__bf16, __bf16>), grid, block, 0, s, (const __bf16*)x, gamma, beta, (__bf16*)y, M, d, d, eps);
    else if (x_dtype == ESMK_BF16 && y_dtype == ESMK_F32)
        hipLaunchKernelGGL((layernorm_kernel<__bf16, float>), grid, block, 0, s, (const __bf16*)x, gamma, beta, (float*)y, M, d, d, eps);
    else
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_layernorm: bad dtype");
    ESMK_CHECK_LAUNCH("esmk_layernorm");
    return ESMK_OK;
}

extern "C" int esmk_add_layernorm(float* x, const void* delta, const float* gamma, const float* beta, void* y,
                                  int y_dtype, int M, int d, float eps, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!delta) return esmk_layernorm(x, ESMK_F32, gamma, beta, y, y_dtype, M, d, eps, stream);
    if (!x || !gamma || !y) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add_layernorm: null pointer");
    if (M < 0 || d <= 0 || (d & 3) || d > LN_MAX_VEC * 256)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_add_layernorm: d=%d must be a multiple of 4 and <= %d", d, LN_MAX_VEC * 256);
    if (M == 0) return ESMK_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    if (y_dtype == ESMK_BF16)
        hipLaunchKernelGGL((layernorm_kernel<float, __bf16, true>), grid, block, 0, s, (const float*)x, gamma, beta, (__bf16*)y, M, d, d, eps, (const __bf16*)delta);
    else if (y_dtype == ESMK_F32)
        hipLaunchKernelGGL((layernorm_kernel<float, float, true>), grid, block, 0, s, (const float*)x, gamma, beta, (float*)y, M, d, d, eps, (const __bf16*)delta);
    else
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add_layernorm: bad dtype");
    ESMK_CHECK_LAUNCH("esmk_add_layernorm");
    return ESMK_OK;
}

extern "C" int esmk_qk_prep(void* qkv, const float* gq, const float* gk, const int32_t* positions, const float* rcos,
                            const float* rsin, int rope_len, int M, int d, int H, float q_scale, float eps,
                            esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!qkv || !gq || !gk || !positions || !rcos || !rsin) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: null pointer");
    if (d != H * 64) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_qk_prep: head dim must be 64 (d=%d, heads=%d)", d, H);
    const int nu = (8 * H + 63) / 64;
    if (nu > QK_MAX_UNITS) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_qk_prep: too many heads (%d)", H);
    if (rope_len <= 0 || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: bad sizes");
    if (M == 0) return ESMK_OK;
    const dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define QK_LAUNCH(NU)                                                                                          \
    hipLaunchKernelGGL(qk_prep_kernel<NU>, grid, block, 0, s, (__bf16*)qkv, gq, gk, positions, rcos, rsin, rope_len, \
                       M, d, H, q_scale, eps)
    switch (nu) {
        case 1: QK_LAUNCH(1); break;
        case 2: QK_LAUNCH(2); break;
        case 3: QK_LAUNCH(3); break;
        case 4: QK_LAUNCH(4); break;
        case 5: QK_LAUNCH(5); break;
        default: QK_LAUNCH(6); break;
    }
#undef QK_LAUNCH
    ESMK_CHECK_LAUNCH("esmk_qk_prep");
    return ESMK_OK;
}

extern "C" int esmk_embed(const int64_t* tokens, const float* table, int vocab, float* out, int M, int d,
                          esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;  // an empty batch is valid and needs no buffers
    if (!tokens || !table || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_embed: null pointer");
    if (d <= 0 || (d & 3) || vocab <= 0 || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_embed: bad sizes");
    if (M == 0) return ESMK_OK;
    const size_t total = (size_t)M * (d >> 2);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                       table, vocab, out, M, d);
    ESMK_CHECK_LAUNCH("esmk_embed");
    return ESMK_OK;
}

extern "C" int esmk_esm3_encode(const esmk_esm3_encoder* enc, const int64_t* sequence, const int64_t* structure,
                                const float* average_plddt, const float* per_res_plddt, const int64_t* ss8,
                                const int64_t* sasa, const int64_t* function_tokens, const int64_t* residue_tokens,
                                float* out, int M, int d, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!enc || !sequence || !structure || !average_plddt || !per_res_plddt || !ss8 || !sasa || !function_tokens ||
        !residue_tokens || !out)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_esm3_encode: null pointer");
    if (d <= 0 || (d & 7) || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_esm3_encode: d must be a multiple of 8");
    if (M == 0) return ESMK_OK;
    const size_t total = (size_t)M * d;
    hipLaunchKernelGGL(esm3_encode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *enc, sequence, structure, average_plddt, per_res_plddt, ss8, sasa, function_tokens,
                       residue_tokens, out, M, d);
    ESMK_CHECK_LAUNCH("esmk_esm3_encode");
    return ESMK_OK;
}
