// Row-wise, HBM-bound kernels: LayerNorm, QK-LayerNorm + rotary, embedding gathers.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace {

// Grid of a persistent row kernel: as many 256-thread blocks as the device holds at once (CUs x occupancy),
// queried once per kernel instantiation, capped by the number of 4-row groups.
template <typename K>
int persistent_blocks(K kernel, size_t lds_bytes, int row_groups, int max_per_cu) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu > max_per_cu) per_cu = max_per_cu;
    return min(row_groups, cus * per_cu);
}

// The grid cap of a kernel instantiation, remembered PER DEVICE (a process may drive several GPUs; threads may race:
// every racer computes the same number, the slots are atomics).
struct DevCap {
    std::atomic<int> v[64];
    template <typename F>
    int get(F compute) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return compute();
        int c = v[dev].load(std::memory_order_relaxed);
        if (c == 0) { c = compute(); v[dev].store(c, std::memory_order_relaxed); }
        return c;
    }
};

// --------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (two-pass mean / variance in fp32).
// Persistent waves: each wave walks rows wave_id, wave_id + n_waves, ... and issues the loads of its NEXT row
// before it reduces / normalises / stores the current one, so a wave always has a whole row in flight
// (a plain one-row-per-wave grid measured 2.9 TB/s on the fp32 -> bf16 pass: latency-bound).  gamma / beta are
// staged once per block in LDS.  NV = float4 vectors per lane (compile time: the row stays in registers).
// --------------------------------------------------------------------------------------------------
constexpr int LN_MAX_VEC = 16;  // 16 float4 per lane -> d <= 4096

template <typename TIN>
struct LnRaw { using type = u32x2; };  // 4 bf16
template <>
struct LnRaw<float> { using type = f32x4; };

__device__ __forceinline__ f32x4 ln_cvt(const f32x4 v) { return v; }
__device__ __forceinline__ f32x4 ln_cvt(const u32x2 p) {
    return f32x4{__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16),
                 __uint_as_float(p.y & 0xffff0000u)};
}

// ADD: 0 = plain LayerNorm; 1 = x (f32, in/out) += delta (bf16) first, updated row written back;
//      2 = normalise x + delta but leave x alone (the delta stays pending for the next pass);
//      3 = x += delta + delta2, written back.  The sum is always formed as (x + delta) + delta2.
template <typename TIN, typename TOUT, int ADD, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TOUT* __restrict__ y, int M,
                                                        int d, int ldy, float eps, const __bf16* __restrict__ delta,
                                                        const __bf16* __restrict__ delta2, float* xw) {
    // xw: where the updated row goes (ADD 1 / 3).  The trunk passes another buffer than x when this pass also
    // captures a hidden state: the residual stream then simply lives on in that buffer (no copy).
    extern __shared__ __attribute__((aligned(16))) float gb[];  // gamma[d] | beta[d]
    using RAW = typename LnRaw<TIN>::type;
    const int lane = threadIdx.x & 63;
    const int nvec = d >> 2;  // d % 4 == 0
    for (int c = threadIdx.x; c < nvec; c += 256) {
        *(f32x4*)(gb + c * 4) = *(const f32x4*)(gamma + c * 4);
        *(f32x4*)(gb + d + c * 4) = beta ? *(const f32x4*)(beta + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    RAW xr[NV];
    u32x2 dr[NV], dr2[NV];
    auto load_row = [&](int r) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                // streamed once: x is next touched a whole sub-block later, delta never again
                xr[i] = __builtin_nontemporal_load((const RAW*)(x + (size_t)r * d + c * 4));
                if constexpr (ADD != 0) dr[i] = __builtin_nontemporal_load((const u32x2*)(delta + (size_t)r * d + c * 4));
                if constexpr (ADD == 3) dr2[i] = __builtin_nontemporal_load((const u32x2*)(delta2 + (size_t)r * d + c * 4));
            }
        }
    };
    if (row < M) load_row(row);
    for (; row < M; row += stride) {
        f32x4 v[NV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nvec) {
                v[i] = ln_cvt(xr[i]);
                if constexpr (ADD != 0) v[i] += ln_cvt(dr[i]);
                if constexpr (ADD == 3) v[i] += ln_cvt(dr2[i]);
                if constexpr (ADD == 1 || ADD == 3)
                    __builtin_nontemporal_store(v[i], (f32x4*)(xw + (size_t)row * d + c * 4));
                sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        if (row + stride < M) load_row(row + stride);  // next row in flight while this one is reduced and stored
        const float mean = wave_sum(sum) / (float)d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                v[i] -= mean;
                sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                const f32x4 o = v[i] * rstd * *(const f32x4*)(gb + c * 4) + *(const f32x4*)(gb + d + c * 4);
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
}

template <typename TIN, typename TOUT, int ADD>
void launch_layernorm(const void* x, const float* gamma, const float* beta, void* y, int M, int d, float eps,
                      const void* delta, hipStream_t s, const void* delta2 = nullptr, float* xw = nullptr) {
    if (!xw) xw = (float*)x;  // in place
    const int nv = (d / 4 + 63) / 64;
    const size_t lds = (size_t)2 * d * sizeof(float);
#define LN_LAUNCH(NV)                                                                                                \
    do {                                                                                                             \
        static DevCap dc; /* resident blocks on the device; measured flat from 2 to 5 per CU, use at most 4 */      \
        const int cap = dc.get([&] { return persistent_blocks(layernorm_kernel<TIN, TOUT, ADD, NV>, lds, 1 << 30, 4); }); \
        hipLaunchKernelGGL((layernorm_kernel<TIN, TOUT, ADD, NV>), dim3(min((M + 3) / 4, cap)), dim3(256), lds, s,   \
                           (const TIN*)x, gamma, beta, (TOUT*)y, M, d, d, eps, (const __bf16*)delta,                 \
                           (const __bf16*)delta2, xw);                                                               \
    } while (0)
    if (nv <= 2) LN_LAUNCH(2);
    else if (nv <= 4) LN_LAUNCH(4);
    else if (nv <= 5) LN_LAUNCH(5);
    else if (nv <= 6) LN_LAUNCH(6);
    else if (nv <= 8) LN_LAUNCH(8);
    else if (nv <= 10) LN_LAUNCH(10);
    else LN_LAUNCH(16);
#undef LN_LAUNCH
}

// --------------------------------------------------------------------------------------------------
// QK-LayerNorm (over the full d_model width, bias-free) + NeoX rotary per 64-wide head, in place.
// One WAVE per token: the q and k rows live in registers, statistics are wave shuffles - no LDS, no block
// barrier.  A "unit" is one 16-byte chunk (8 columns) of the token's contiguous [q | k] row: lane L owns chunks
// L, L+64, ... so every load / store instruction of the wave covers 1 KiB of contiguous memory.  The rotary
// partner of a column (j <-> j+32 inside a head) lives 4 chunks away = in lane L^4 of the same unit.
// Persistent waves with the next token's row in flight (see layernorm_kernel).
// --------------------------------------------------------------------------------------------------
constexpr int QK_MAX_UNITS = 12;  // per lane -> 16*H <= 768 -> H <= 48

// value of lane ^ 4 by two bank-masked DPP moves (rows of 16 lanes = 4 banks of 4: banks 0 / 2 take lane + 4, banks
// 1 / 3 lane - 4) - vector ALU only, where __shfl_xor goes through the LDS crossbar (ds_bpermute + wait)
__device__ __forceinline__ float lane_xor4(float v) {
    const int x = __float_as_int(v);
    int r = __builtin_amdgcn_update_dpp(0, x, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, x, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
    return __int_as_float(r);
}

__device__ __forceinline__ void unpack8(const u32x4 p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(p[e] << 16);
        f[2 * e + 1] = __uint_as_float(p[e] & 0xffff0000u);
    }
}

// LN = false (qk_layernorm=False block variant, attention.py:34-38): rotary and the q scale only.
template <int NU, bool LN = true>  // units per lane (compile-time so the row stays in registers); NU = ceil(16*H / 64)
__global__ __launch_bounds__(256) void qk_prep_kernel(__bf16* __restrict__ qkv, const float* __restrict__ gq,
                                                      const float* __restrict__ gk, const float* __restrict__ bq,
                                                      const float* __restrict__ bk,
                                                      const int* __restrict__ positions,
                                                      const float* __restrict__ rcos,
                                                      const float* __restrict__ rsin, int rope_len, int M, int d,
                                                      int H, float q_scale, float eps) {
    const int lane = threadIdx.x & 63;
    const int units_half = 8 * H;  // chunks of q; same count for k
    const int nunits = 2 * units_half;
    const int stride = gridDim.x * 4;
    int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    // Only the LAST unit of a lane can fall outside the row (NU = ceil(nunits/64)): its lanes read chunk 0 again
    // and are masked out of the statistics and the store, so that no other unit needs a per-lane guard.
    const bool last_on = lane + (NU - 1) * 64 < nunits;
    const float last_w = last_on ? 1.0f : 0.0f;
    int uoff[NU];   // element offset of the unit inside the token's [q | k] row
    float kw[NU];   // 1.0 for a k chunk, 0.0 for a q chunk
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int unit = lane + u * 64;
        const bool on = (u < NU - 1) || last_on;
        uoff[u] = on ? unit * 8 : 0;
        kw[u] = (on && unit >= units_half) ? 1.0f : 0.0f;
    }
    u32x4 raw[NU];
    int pos_next = 0;
    auto load_tok = [&](int t) {
        const __bf16* row = qkv + (size_t)t * 3 * d;
#pragma unroll
        for (int u = 0; u < NU; ++u) raw[u] = *(const u32x4*)(row + uoff[u]);
        pos_next = positions[t];
    };
    if (tok < M) load_tok(tok);
    const float inv_d = 1.0f / (float)d;
    const int j0 = (lane & 3) * 8;                 // rotary frequency index of this lane's element 0
    const float sgn = (lane & 4) ? 1.0f : -1.0f;   // first half: x1*c - x2*s; second half: x2*c + x1*s
    for (; tok < M; tok += stride) {
        __bf16* row = qkv + (size_t)tok * 3 * d;
        float v[NU][8];
        float st = 0.f, sk = 0.f;  // sum over q and k together, and over k alone
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            unpack8(raw[u], v[u]);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[u][e];
            if (u == NU - 1) s *= last_w;
            st += s;
            sk += s * kw[u];
        }
        int pos = pos_next;
        if (tok + stride < M) load_tok(tok + stride);
        float rstd_k = 1.0f, rstd_q = q_scale;
        if constexpr (LN) {
            st = wave_sum(st);
            sk = wave_sum(sk);
            const float mean_k = sk * inv_d, mean_q = (st - sk) * inv_d;
            st = 0.f; sk = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float m = kw[u] != 0.f ? mean_k : mean_q;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[u][e] -= m;
                    s += v[u][e] * v[u][e];
                }
                if (u == NU - 1) s *= last_w;
                st += s;
                sk += s * kw[u];
            }
            st = wave_sum(st);
            sk = wave_sum(sk);
            rstd_k = rsqrtf(sk * inv_d + eps);
            rstd_q = rsqrtf((st - sk) * inv_d + eps) * q_scale;
        }
        pos = pos < rope_len ? pos : rope_len - 1;  // host validates; clamp keeps the load in bounds
        const float* cs = rcos + (size_t)pos * 32 + j0;
        const float* sn = rsin + (size_t)pos * 32 + j0;
        const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4);
        const f32x4 s0 = *(const f32x4*)sn * sgn, s1 = *(const f32x4*)(sn + 4) * sgn;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            // every lane takes part in the exchange: the partner lane L^4 is in range whenever L is (nunits % 8 == 0)
            const bool isk = kw[u] != 0.f;
            const float rs = isk ? rstd_k : rstd_q;
            f32x4 ga = f32x4{rs, rs, rs, rs}, gb = ga;
            f32x4 ba = f32x4{0.f, 0.f, 0.f, 0.f}, bb = ba;  // biased q_ln / k_ln (bias=True blocks): beta, q's scaled like q
            if constexpr (LN) {
                const float* g = (isk ? gk - units_half * 8 : gq) + uoff[u];
                ga = *(const f32x4*)g * rs;
                gb = *(const f32x4*)(g + 4) * rs;
                if (bq) {  // kernel argument: uniform
                    const float* b = (isk ? bk - units_half * 8 : bq) + uoff[u];
                    const float bs = isk ? 1.0f : q_scale;
                    ba = *(const f32x4*)b * bs;
                    bb = *(const f32x4*)(b + 4) * bs;
                }
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = v[u][e] * (e < 4 ? ga[e & 3] : gb[e & 3]) + (e < 4 ? ba[e & 3] : bb[e & 3]);
                const float pn = lane_xor4(n);
                o[e] = n * (e < 4 ? c0[e & 3] : c1[e & 3]) + pn * (e < 4 ? s0[e & 3] : s1[e & 3]);
            }
            if (u < NU - 1 || last_on)
                *(u32x4*)(row + uoff[u]) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                 pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        }
    }
}

// --------------------------------------------------------------------------------------------------
// The k half alone (esmk_k_prep; q is normalised and rotated by the attention kernel's prologue): k_ln over the d_model
// columns of k + rotary, in place.  A k row is half the bytes of the [q | k] row above, so one HALF-wave owns a token
// (two tokens per wave instruction stream: the per-token latency chain - loads, two reductions, table rows - is
// amortised as before); reductions are five xor steps inside the half.  Lane l of the half owns chunks l, l + 32, ...
// of its token's k row; the rotary partner is still lane ^ 4.  With qstats, lanes p < parts of the half also reduce the
// token's q partials (esmk_gemm_qkv_stats) to (mean, rstd) like parallel variances (each partial's cancellation stays
// its own).  Measured at 32768 x 1152 (tools/kprep_ab.py, k from HBM): 42 us = 151 MB at 3.6 TB/s; without the store 30,
// without the table rows 37, without the q statistics 39; flat from 2 to 8 resident blocks per CU, and a whole wave per
// token measured the same - the pass touches one third of every [q | k | v] row, and reads 2304-byte pieces 6912 bytes
// apart slower than the [q | k] pass reads its 4608 (5.0 TB/s) or the LayerNorm passes their contiguous rows (6.3).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_sum(float v) {  // sum over the lane's 32-lane half, in every lane of it
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xb1>{});   // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp(v, std::integral_constant<int, 0x4e>{});   // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other quad of the 8
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: the other 8 of the row
    return v + __shfl_xor(v, 16);
}

template <int NU>  // chunks per lane: ceil(8*H / 32)
__global__ __launch_bounds__(256) void k_prep_kernel(__bf16* __restrict__ qkv, const float* __restrict__ gk,
                                                     const float* __restrict__ bk, const int* __restrict__ positions,
                                                     const float* __restrict__ rcos, const float* __restrict__ rsin,
                                                     int rope_len, int M, int d, int H, float eps,
                                                     const float* __restrict__ qstats, int parts,
                                                     float* __restrict__ q_mean_rstd) {
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    const int nunits = 8 * H;
    const int stride = gridDim.x * 8;
    int tok = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool last_on = l32 + (NU - 1) * 32 < nunits;  // only the last chunk of a lane can fall outside the row
    const float last_w = last_on ? 1.0f : 0.0f;
    int uoff[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) uoff[u] = ((u < NU - 1) || last_on) ? (l32 + u * 32) * 8 : 0;
    u32x4 raw[NU];
    int pos_next = 0;
    f32x2 part_next = f32x2{0.f, 0.f};
    // a half past the end re-reads the last token and stores nothing (it still takes part in every exchange)
    auto load_tok = [&](int t) {
        const int tc = t < M ? t : M - 1;
        const __bf16* row = qkv + (size_t)tc * 3 * d + d;
#pragma unroll
        for (int u = 0; u < NU; ++u) raw[u] = __builtin_nontemporal_load((const u32x4*)(row + uoff[u]));
        pos_next = positions[tc];
        if (qstats && l32 < parts) part_next = *(const f32x2*)(qstats + ((size_t)tc * parts + l32) * 2);
    };
    const int tok_w = tok - (lane >> 5);  // first token of the wave: the loop condition is wave-uniform
    if (tok_w < M) load_tok(tok);
    const float inv_d = 1.0f / (float)d;
    const int j0 = (lane & 3) * 8;
    const float sgn = (lane & 4) ? 1.0f : -1.0f;
    for (int tw = tok_w; tw < M; tw += stride, tok += stride) {
        const bool live = tok < M;
        __bf16* row = qkv + (size_t)(live ? tok : M - 1) * 3 * d + d;
        float v[NU][8];
        float st = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            unpack8(raw[u], v[u]);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[u][e];
            if (u == NU - 1) s *= last_w;
            st += s;
        }
        int pos = pos_next;
        const f32x2 part = part_next;
        if (tw + stride < M) load_tok(tok + stride);
        if (qstats) {  // kernel argument: uniform
            const float n_i = (float)d / (float)parts;
            const float on = l32 < parts ? 1.0f : 0.0f;
            const float mean_q = half_sum(part[0] * on) * inv_d;
            const float mi = part[0] / n_i;
            const float m2 = half_sum(on * ((part[1] - part[0] * mi) + n_i * (mi - mean_q) * (mi - mean_q)));
            if (l32 == 0 && live) *(f32x2*)(q_mean_rstd + (size_t)tok * 2) = f32x2{mean_q, rsqrtf(fmaxf(m2 * inv_d, 0.f) + eps)};
        }
        const float mean = half_sum(st) * inv_d;
        st = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[u][e] -= mean;
                s += v[u][e] * v[u][e];
            }
            if (u == NU - 1) s *= last_w;
            st += s;
        }
        const float rstd = rsqrtf(half_sum(st) * inv_d + eps);
        pos = pos < rope_len ? pos : rope_len - 1;  // host validates; clamp keeps the load in bounds
        const float* cs = rcos + (size_t)pos * 32 + j0;
        const float* sn = rsin + (size_t)pos * 32 + j0;
        const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4);
        const f32x4 s0 = *(const f32x4*)sn * sgn, s1 = *(const f32x4*)(sn + 4) * sgn;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float* g = gk + uoff[u];
            const f32x4 ga = *(const f32x4*)g * rstd, gb = *(const f32x4*)(g + 4) * rstd;
            f32x4 ba = f32x4{0.f, 0.f, 0.f, 0.f}, bb = ba;
            if (bk) {  // kernel argument: uniform
                ba = *(const f32x4*)(bk + uoff[u]);
                bb = *(const f32x4*)(bk + uoff[u] + 4);
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = v[u][e] * (e < 4 ? ga[e & 3] : gb[e & 3]) + (e < 4 ? ba[e & 3] : bb[e & 3]);
                const float pn = lane_xor4(n);
                o[e] = n * (e < 4 ? c0[e & 3] : c1[e & 3]) + pn * (e < 4 ? s0[e & 3] : s1[e & 3]);
            }
            if (live && (u < NU - 1 || last_on))
                *(u32x4*)(row + uoff[u]) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                 pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
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

// Mean over the rows of every segment: block = (segment, 64-column slab); 16 row groups x 16 lanes x float4,
// rows strided by 16, row groups combined through LDS.  Empty segments give zeros.
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, int ldx,
                                                           const int* __restrict__ cu, float* __restrict__ out,
                                                           int d) {
    __shared__ f32x4 part[16][16];
    const int seg = blockIdx.x;
    const int c = blockIdx.y * 64 + (threadIdx.x & 15) * 4;
    const int rg = threadIdx.x >> 4;
    const int r0 = cu[seg], r1 = cu[seg + 1];
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < d)
        for (int r = r0 + rg; r < r1; r += 16) acc += *(const f32x4*)(x + (size_t)r * ldx + c);
    part[rg][threadIdx.x & 15] = acc;
    __syncthreads();
    if (rg == 0 && c < d) {
#pragma unroll
        for (int k = 1; k < 16; ++k) acc += part[k][threadIdx.x];
        const float inv = r1 > r0 ? 1.0f / (float)(r1 - r0) : 0.f;
        *(f32x4*)(out + (size_t)seg * d + c) = acc * inv;
    }
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

// esm3_encode with the reference's default filling and structure-token fix-ups done here (esm3.py:317-359): a NULL
// track takes its default, a track with rows == 1 is broadcast over the batch (the reference builds its defaults with
// batch 1 and lets the sum broadcast), structure tokens are forced at BOS / PAD / EOS / chain-break positions of the
// sequence track, -1 means MASK.  Out-of-range ids are clamped and counted.
__global__ void esm3_encode_ex_kernel(esmk_esm3_encoder enc, esmk_esm3_inputs in, float* __restrict__ out, int B, int L,
                                      int d, int* __restrict__ err) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = B * L;
    if (idx >= (size_t)M * d) return;
    const int t = (int)(idx / d), c = (int)(idx - (size_t)t * d);
    const int b = t / L, l = t - b * L;
    auto at = [&](int track) -> size_t { return (size_t)(in.rows[track] == 1 ? 0 : b) * L + l; };
    auto oob = [](long long v, int n) -> int { return v < 0 || v >= n; };
    const long long sq_raw = in.sequence ? in.sequence[at(0)] : 32;  // default: the sequence MASK token
    const long long sq = clampi(sq_raw, 64);
    long long st_raw = in.structure ? in.structure[at(1)] : 4096;
    if (st_raw == -1) st_raw = 4096;                   // STRUCTURE_MASK_TOKEN
    if (sq_raw == 0) st_raw = 4098;                    // BOS
    else if (sq_raw == 1) st_raw = 4099;               // PAD
    else if (sq_raw == 2) st_raw = 4097;               // EOS
    else if (sq_raw == 31) st_raw = 4100;              // chain break
    const long long stv = clampi(st_raw, 4101);
    const long long s8_raw = in.ss8 ? in.ss8[at(2)] : 0, sa_raw = in.sasa ? in.sasa[at(3)] : 0;
    const long long s8 = clampi(s8_raw, 11), sa = clampi(sa_raw, 19);
    float acc = enc.sequence_embed[sq * d + c];
    {
        const float v = in.average_plddt ? in.average_plddt[at(6)] : 1.0f;
        float s = enc.plddt_b[c];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float z = (v - (float)k * (1.0f / 15.0f)) * 16.0f;
            s += expf(-z * z) * enc.plddt_w[c * 16 + k];
        }
        acc += s;
    }
    {
        const float v = in.per_res_plddt ? in.per_res_plddt[at(7)] : 0.0f;
        float s = enc.res_plddt_b[c];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float z = (v - (float)k * (1.0f / 15.0f)) * 16.0f;
            s += expf(-z * z) * enc.res_plddt_w[c * 16 + k];
        }
        acc += s;
    }
    acc += enc.structure_embed[stv * d + c];
    acc += enc.ss8_embed[s8 * d + c];
    acc += enc.sasa_embed[sa * d + c];
    if (in.function_tokens) {
        const int d8 = d >> 3;
        const int track = c / d8, cc = c - track * d8;
        const long long f = clampi(in.function_tokens[at(4) * 8 + track], 260);
        if (f != 0) acc += enc.function_embed[((size_t)track * 260 + f) * d8 + cc];
    }
    if (in.residue_tokens) {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const long long a = clampi(in.residue_tokens[at(5) * 16 + r], 1478);
            if (a != 0) s += enc.residue_embed[a * d + c];
        }
        acc += s;
    }
    out[idx] = acc;
    if (err && c == 0) {  // the thread of channel 0 checks every id of its token
        int bad = oob(sq_raw, 64) + oob(st_raw, 4101) + oob(s8_raw, 11) + oob(sa_raw, 19);
        if (in.function_tokens)
            for (int k = 0; k < 8; ++k) bad += oob(in.function_tokens[at(4) * 8 + k], 260);
        if (in.residue_tokens)
            for (int r = 0; r < 16; ++r) bad += oob(in.residue_tokens[at(5) * 16 + r], 1478);
        if (bad) atomicAdd(err, bad);
    }
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
    if (x_dtype == ESMK_F32 && y_dtype == ESMK_BF16)
        launch_layernorm<float, __bf16, 0>(x, gamma, beta, y, M, d, eps, nullptr, s);
    else if (x_dtype == ESMK_F32 && y_dtype == ESMK_F32)
        launch_layernorm<float, float, 0>(x, gamma, beta, y, M, d, eps, nullptr, s);
    else if (x_dtype == ESMK_BF16 && y_dtype == ESMK_BF16)
        launch_layernorm<__bf16, __bf16, 0>(x, gamma, beta, y, M, d, eps, nullptr, s);
    else if (x_dtype == ESMK_BF16 && y_dtype == ESMK_F32)
        launch_layernorm<__bf16, float, 0>(x, gamma, beta, y, M, d, eps, nullptr, s);
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
    if (y_dtype == ESMK_BF16)
        launch_layernorm<float, __bf16, 1>(x, gamma, beta, y, M, d, eps, delta, s);
    else if (y_dtype == ESMK_F32)
        launch_layernorm<float, float, 1>(x, gamma, beta, y, M, d, eps, delta, s);
    else
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add_layernorm: bad dtype");
    ESMK_CHECK_LAUNCH("esmk_add_layernorm");
    return ESMK_OK;
}

// Internal form used by the trunk: x_out = x_in + delta_a + delta_b (both deltas required), y = LN(x_out).  x_out may be
// another buffer than x_in (hidden-state capture without a copy).
int esmk_add2_layernorm_io(const float* x_in, float* x_out, const void* delta_a, const void* delta_b, const float* gamma,
                           const float* beta, void* y, int y_dtype, int M, int d, float eps, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!x_in || !x_out || !delta_a || !delta_b || !gamma || !y) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add2_layernorm: null pointer");
    if (M < 0 || d <= 0 || (d & 3) || d > LN_MAX_VEC * 256)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_add2_layernorm: d=%d must be a multiple of 4 and <= %d", d, LN_MAX_VEC * 256);
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == ESMK_BF16) launch_layernorm<float, __bf16, 3>(x_in, gamma, beta, y, M, d, eps, delta_a, s, delta_b, x_out);
    else if (y_dtype == ESMK_F32) launch_layernorm<float, float, 3>(x_in, gamma, beta, y, M, d, eps, delta_a, s, delta_b, x_out);
    else ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add2_layernorm: bad dtype");
    ESMK_CHECK_LAUNCH("esmk_add2_layernorm");
    return ESMK_OK;
}

extern "C" int esmk_add2_layernorm(float* x, const void* delta_a, const void* delta_b, int write_x, const float* gamma,
                                   const float* beta, void* y, int y_dtype, int M, int d, float eps,
                                   esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!delta_b && write_x) return esmk_add_layernorm(x, delta_a, gamma, beta, y, y_dtype, M, d, eps, stream);
    if (!x || !delta_a || !gamma || !y) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add2_layernorm: null pointer");
    if (delta_b && !write_x) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_add2_layernorm: two deltas must be written back");
    if (M < 0 || d <= 0 || (d & 3) || d > LN_MAX_VEC * 256)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_add2_layernorm: d=%d must be a multiple of 4 and <= %d", d, LN_MAX_VEC * 256);
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == ESMK_BF16) {
        if (delta_b) launch_layernorm<float, __bf16, 3>(x, gamma, beta, y, M, d, eps, delta_a, s, delta_b);
        else launch_layernorm<float, __bf16, 2>(x, gamma, beta, y, M, d, eps, delta_a, s);
    } else if (y_dtype == ESMK_F32) {
        if (delta_b) launch_layernorm<float, float, 3>(x, gamma, beta, y, M, d, eps, delta_a, s, delta_b);
        else launch_layernorm<float, float, 2>(x, gamma, beta, y, M, d, eps, delta_a, s);
    } else {
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_add2_layernorm: bad dtype");
    }
    ESMK_CHECK_LAUNCH("esmk_add2_layernorm");
    return ESMK_OK;
}

extern "C" int esmk_qk_prep(void* qkv, const float* gq, const float* gk, const int32_t* positions, const float* rcos,
                            const float* rsin, int rope_len, int M, int d, int H, float q_scale, float eps,
                            esmk_stream_t stream) {
    return esmk_qk_prep_ex(qkv, gq, gk, nullptr, nullptr, positions, rcos, rsin, rope_len, M, d, H, q_scale, eps, stream);
}

extern "C" int esmk_qk_prep_ex(void* qkv, const float* gq, const float* gk, const float* bq, const float* bk,
                               const int32_t* positions, const float* rcos, const float* rsin, int rope_len, int M,
                               int d, int H, float q_scale, float eps, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!qkv || !positions || !rcos || !rsin) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: null pointer");
    if ((gq == nullptr) != (gk == nullptr))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: gamma_q and gamma_k must both be given (QK-LayerNorm) or both be NULL");
    if ((bq == nullptr) != (bk == nullptr) || (bq && !gq))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: beta_q and beta_k go together and need the gammas");
    const bool ln = gq != nullptr;
    if (d != H * 64) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_qk_prep: head dim must be 64 (d=%d, heads=%d)", d, H);
    const int nu = (16 * H + 63) / 64;
    if (nu > QK_MAX_UNITS) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_qk_prep: too many heads (%d)", H);
    if (rope_len <= 0 || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_qk_prep: bad sizes");
    if (M == 0) return ESMK_OK;
    const dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
    /* grid = exactly the blocks the device holds at once (3 per CU for 18 heads): measured best */
#define QK_LAUNCH(NU)                                                                                          \
    do {                                                                                                       \
        static DevCap dc;                                                                                      \
        const int cap = dc.get([&] { return persistent_blocks(qk_prep_kernel<NU>, 0, 1 << 30, 8); });          \
        if (ln)                                                                                                \
            hipLaunchKernelGGL((qk_prep_kernel<NU, true>), dim3(min((M + 3) / 4, cap)), block, 0, s, (__bf16*)qkv, gq, \
                               gk, bq, bk, positions, rcos, rsin, rope_len, M, d, H, q_scale, eps); \
        else                                                                                                   \
            hipLaunchKernelGGL((qk_prep_kernel<NU, false>), dim3(min((M + 3) / 4, cap)), block, 0, s, (__bf16*)qkv, \
                               gq, gk, bq, bk, positions, rcos, rsin, rope_len, M, d, H, q_scale, eps); \
    } while (0)
    switch (nu) {
        case 1: QK_LAUNCH(1); break;
        case 2: QK_LAUNCH(2); break;
        case 3: QK_LAUNCH(3); break;
        case 4: QK_LAUNCH(4); break;
        case 5: QK_LAUNCH(5); break;
        case 6: QK_LAUNCH(6); break;
        case 7: case 8: QK_LAUNCH(8); break;
        case 9: case 10: QK_LAUNCH(10); break;
        default: QK_LAUNCH(12); break;
    }
#undef QK_LAUNCH
    ESMK_CHECK_LAUNCH("esmk_qk_prep");
    return ESMK_OK;
}

// The k half alone: q is normalised and rotated inside the attention kernel (esmk_attn_varlen_qfused).
extern "C" int esmk_k_prep(void* qkv, const float* gk, const float* bk, const int32_t* positions, const float* rcos,
                           const float* rsin, int rope_len, int M, int d, int H, float eps, const float* qstats, int parts,
                           float* q_mean_rstd, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!qkv || !gk || !positions || !rcos || !rsin) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_k_prep: null pointer");
    if (qstats && (!q_mean_rstd || parts < 1 || parts > 32 || (d % parts) || ((uintptr_t)qstats & 7) || ((uintptr_t)q_mean_rstd & 7)))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_k_prep: qstats needs q_mean_rstd (both 8-byte aligned) and 1 <= parts <= 32 dividing d");
    if (d != H * 64) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_k_prep: head dim must be 64 (d=%d, heads=%d)", d, H);
    const int nu = (8 * H + 31) / 32;
    if (nu > 12) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_k_prep: too many heads (%d)", H);
    if (rope_len <= 0 || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_k_prep: bad sizes");
    const dim3 block(256);
    hipStream_t s = (hipStream_t)stream;
#define K_LAUNCH(NU)                                                                                           \
    do {                                                                                                       \
        static DevCap dc;                                                                                      \
        const int cap = dc.get([&] { return persistent_blocks(k_prep_kernel<NU>, 0, 1 << 30, 8); });  /* flat from 2 to 8 per CU */ \
        hipLaunchKernelGGL((k_prep_kernel<NU>), dim3(min((M + 7) / 8, cap)), block, 0, s, (__bf16*)qkv, gk, bk,  \
                           positions, rcos, rsin, rope_len, M, d, H, eps, qstats, parts, q_mean_rstd);          \
    } while (0)
    switch (nu) {
        case 1: K_LAUNCH(1); break;
        case 2: K_LAUNCH(2); break;
        case 3: K_LAUNCH(3); break;
        case 4: K_LAUNCH(4); break;
        case 5: K_LAUNCH(5); break;
        case 6: K_LAUNCH(6); break;
        case 7: case 8: K_LAUNCH(8); break;
        case 9: case 10: K_LAUNCH(10); break;
        default: K_LAUNCH(12); break;
    }
#undef K_LAUNCH
    ESMK_CHECK_LAUNCH("esmk_k_prep");
    return ESMK_OK;
}

// Device-side batch description of a right-padded [B, L] token matrix (esmk_segments_from_tokens): one block per row.
__global__ __launch_bounds__(256) void segments_from_tokens_kernel(const int64_t* __restrict__ tokens,
                                                                   const uint8_t* __restrict__ mask, int L, long long pad_id,
                                                                   int vocab, int B, int* __restrict__ cu,
                                                                   int* __restrict__ err) {
    __shared__ int s_len, s_bad;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { s_len = L; s_bad = 0; }
    __syncthreads();
    int first_pad = L, bad = 0;  // valid prefix = the leading run of real tokens (what the header documents)
    const int64_t* row = tokens + (size_t)b * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const long long t = row[i];
        const bool v = mask ? mask[(size_t)b * L + i] != 0 : t != pad_id;
        if (!v) first_pad = min(first_pad, i);
        if (v && i > 0) {  // a real token behind a padding position: the row is not right-padded
            const bool pv = mask ? mask[(size_t)b * L + i - 1] != 0 : row[i - 1] != pad_id;
            if (!pv) ++bad;
        }
        if (t < 0 || t >= vocab) ++bad;
    }
    if (first_pad < L) atomicMin(&s_len, first_pad);
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        cu[2 * b] = b * L;
        cu[2 * b + 1] = b * L + s_len;
        if (b == B - 1) cu[2 * B] = B * L;
        if (s_bad && err) atomicAdd(err, s_bad);
    }
}

extern "C" int esmk_segments_from_tokens(const int64_t* tokens, const uint8_t* mask, int B, int L, int64_t pad_id,
                                         int vocab, int32_t* cu_seqlens, int32_t* err_count, esmk_stream_t stream) {
    if (B == 0) return ESMK_OK;
    if (!tokens || !cu_seqlens) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_segments_from_tokens: null pointer");
    if (B < 0 || L <= 0 || vocab <= 0 || (long long)B * L > 0x7fffffffLL)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_segments_from_tokens: bad sizes B=%d L=%d", B, L);
    hipLaunchKernelGGL(segments_from_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tokens, mask, L,
                       (long long)pad_id, vocab, B, cu_seqlens, err_count);
    ESMK_CHECK_LAUNCH("esmk_segments_from_tokens");
    return ESMK_OK;
}

// Device-side batch description of a [B, L] token matrix with ANY padding pattern (reference semantics of
// `sequence_id = tokens != pad`, esmc.py:133-135 with attention.py:65-72: the non-pad tokens of a row attend to each
// other wherever they sit, rotary position = column).  One block per row: the non-pad tokens are moved to the front of
// the row's L slots in their original order (a no-op for right-padded rows), the pad tokens behind them.
//   cu[2b], cu[2b+1], cu[2B]   segment bounds: valid tokens of row b, then its padding slots
//   perm[b*L + j]              flat source index (b*L + column) of the token now in slot j
//   positions[b*L + j]         its column = rotary position
//   tok_out[b*L + j]           the token itself
__global__ __launch_bounds__(256) void compact_tokens_kernel(const int64_t* __restrict__ tokens, int L, long long pad_id,
                                                             int vocab, int B, int* __restrict__ cu, int* __restrict__ perm,
                                                             int* __restrict__ positions, int64_t* __restrict__ tok_out,
                                                             int* __restrict__ err) {
    __shared__ int s_wave[4], s_total, s_bad;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* row = tokens + (size_t)b * L;
    if (tid == 0) { s_total = 0; s_bad = 0; }
    __syncthreads();
    int cnt = 0, bad = 0;
    for (int i = tid; i < L; i += 256) {
        const long long t = row[i];
        cnt += t != pad_id;
        bad += (t < 0 || t >= vocab);
    }
    cnt = (int)wave_sum((float)cnt);  // exact: counts stay far below 2^24
    if (lane == 0 && cnt) atomicAdd(&s_total, cnt);
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    const int n_valid = s_total;
    int base_v = 0, base_p = 0;  // valid / pad tokens in the chunks already placed
    for (int i0 = 0; i0 < L; i0 += 256) {
        const int i = i0 + tid;
        const long long t = i < L ? row[i] : pad_id;
        const bool in = i < L, v = in && t != pad_id;
        const unsigned long long m = __ballot(v);
        const int rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = 0, chunk_v = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? s_wave[w] : 0;
            chunk_v += s_wave[w];
        }
        if (in) {
            const int vr = base_v + before + rank_in_wave;              // rank among the row's valid tokens
            const int j = v ? vr : n_valid + (i - vr);                  // pads: their rank among the pads
            const size_t o = (size_t)b * L + j;
            perm[o] = b * L + i;
            positions[o] = i;
            tok_out[o] = t;
        }
        base_v += chunk_v;
        __syncthreads();
    }
    if (tid == 0) {
        cu[2 * b] = b * L;
        cu[2 * b + 1] = b * L + n_valid;
        if (b == B - 1) cu[2 * B] = B * L;
        if (s_bad && err) atomicAdd(err, s_bad);
    }
}

// Inverse of the compaction, in place, on n_mat matrices [B*L, w] f32: row slot j of batch row b goes back to its source
// column, padding columns become zero (the reference's pad_input, esmc.py:159-164).  A thread owns one float4 column
// and walks the valid slots from the last to the first: a token only ever moves right (source column >= slot), so the
// slot it lands on has already been read.  Right-padded rows move nothing.
__global__ __launch_bounds__(256) void unpack_rows_kernel(float* __restrict__ x, const int* __restrict__ perm,
                                                          const int* __restrict__ cu, int L, int w, size_t mat_stride) {
    const int b = blockIdx.x;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c >= w) return;
    float* base = x + (size_t)blockIdx.z * mat_stride;
    const int r0 = b * L, n_valid = cu[2 * b + 1] - r0;
    for (int j = n_valid - 1; j >= 0; --j) {
        const int src = perm[r0 + j];
        if (src != r0 + j) *(f32x4*)(base + (size_t)src * w + c) = *(const f32x4*)(base + (size_t)(r0 + j) * w + c);
    }
    for (int j = n_valid; j < L; ++j) *(f32x4*)(base + (size_t)perm[r0 + j] * w + c) = f32x4{0.f, 0.f, 0.f, 0.f};
}

extern "C" int esmk_compact_tokens(const int64_t* tokens, int B, int L, int64_t pad_id, int vocab, int32_t* cu_seqlens,
                                   int32_t* perm, int32_t* positions, int64_t* tokens_out, int32_t* err_count,
                                   esmk_stream_t stream) {
    if (B == 0) return ESMK_OK;
    if (!tokens || !cu_seqlens || !perm || !positions || !tokens_out)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_compact_tokens: null pointer");
    if (B < 0 || L <= 0 || vocab <= 0 || (long long)B * L > 0x7fffffffLL)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_compact_tokens: bad sizes B=%d L=%d", B, L);
    if (tokens == tokens_out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_compact_tokens: tokens_out must not alias tokens");
    hipLaunchKernelGGL(compact_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tokens, L, (long long)pad_id, vocab,
                       B, cu_seqlens, perm, positions, tokens_out, err_count);
    ESMK_CHECK_LAUNCH("esmk_compact_tokens");
    return ESMK_OK;
}

extern "C" int esmk_unpack_rows(float* x, int n_mat, int B, int L, int w, const int32_t* perm, const int32_t* cu_seqlens,
                                esmk_stream_t stream) {
    if (B == 0 || n_mat == 0) return ESMK_OK;
    if (!x || !perm || !cu_seqlens) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_unpack_rows: null pointer");
    if (B < 0 || n_mat < 0 || n_mat > 65535 || L <= 0 || w <= 0 || (w & 3) || ((uintptr_t)x & 15))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_unpack_rows: bad sizes (w must be a multiple of 4, x 16-byte aligned)");
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(B, (w / 4 + 255) / 256, n_mat), dim3(256), 0, (hipStream_t)stream, x, perm,
                       cu_seqlens, L, w, (size_t)B * L * w);
    ESMK_CHECK_LAUNCH("esmk_unpack_rows");
    return ESMK_OK;
}

extern "C" int esmk_segment_mean(const float* x, int ldx, const int32_t* cu_seqlens, int n_segments, float* out, int d,
                                 esmk_stream_t stream) {
    if (n_segments == 0) return ESMK_OK;
    if (!x || !cu_seqlens || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_segment_mean: null pointer");
    if (n_segments < 0 || d <= 0 || (d & 3) || ldx < d || (ldx & 3))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_segment_mean: d and ldx must be multiples of 4, ldx >= d");
    hipLaunchKernelGGL(segment_mean_kernel, dim3(n_segments, (d + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       cu_seqlens, out, d);
    ESMK_CHECK_LAUNCH("esmk_segment_mean");
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

extern "C" int esmk_esm3_encode_ex(const esmk_esm3_encoder* enc, const esmk_esm3_inputs* in, float* out, int B, int L, int d,
                                   int32_t* err_count, esmk_stream_t stream) {
    if (B == 0 || L == 0) return ESMK_OK;
    if (!enc || !in || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_esm3_encode_ex: null pointer");
    if (d <= 0 || (d & 7) || B < 0 || L < 0 || (long long)B * L > 0x7fffffffLL)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_esm3_encode_ex: bad sizes (d must be a multiple of 8)");
    const void* ptrs[8] = {in->sequence, in->structure, in->ss8, in->sasa, in->function_tokens, in->residue_tokens,
                           in->average_plddt, in->per_res_plddt};
    for (int i = 0; i < 8; ++i)
        if (ptrs[i] && in->rows[i] != 1 && in->rows[i] != B)
            ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_esm3_encode_ex: track %d has %d batch rows, expected 1 or %d", i, in->rows[i], B);
    const size_t total = (size_t)B * L * d;
    hipLaunchKernelGGL(esm3_encode_ex_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *enc,
                       *in, out, B, L, d, err_count);
    ESMK_CHECK_LAUNCH("esmk_esm3_encode_ex");
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
