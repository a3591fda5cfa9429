// Small-batch (latency) GEMMs of the trunk: a few hundred to a few thousand tokens, where every weight byte comes from
// HBM exactly once and a K-step of the big-tile kernels costs a full memory round trip.
//
// One all-LDS-DMA kernel (smgemm_kernel) on v_mfma_f32_16x16x32_bf16 with the orientation of gemm.hip (MFMA rows = output
// channels, MFMA columns = tokens, LDS rows of 128 B with the 16-byte chunk XOR-swizzled by (row>>1)&7), a ring of NS
// stages with NS-1 K-steps in flight, counted `s_waitcnt vmcnt` and a RAW `s_barrier` per K-step (the fence inside
// __syncthreads() is `s_waitcnt vmcnt(0)`: it would drain the ring at every step), in two roles:
//
//   LayerNorm-folded projection   out = epi(LayerNorm(x) . W^T)
//       (reference esm/layers/attention.py:26-28,49 layernorm_qkv; esm/layers/blocks.py:30-33 swiglu_ln_ffn;
//       esm/layers/geom_attention.py:65-72).  LayerNorm is affine per row, so with W' = W . diag(gamma),
//       u[n] = sum_k W'[n,k] and v[n] = sum_k W[n,k] beta[k]:
//           LN(x) . W^T  =  rstd_t * (x . W'^T  -  mean_t * u)  +  v
//       The kernel multiplies the RAW residual stream (its bf16 copy xb) with the pre-folded weight W' - every operand
//       byte goes global -> LDS by DMA, no pass over x, no per-element work in the K loop - and applies mean_t / rstd_t
//       (prologue: from the per-row partial sums the producer of x left behind) and u / v in the epilogue.  x is
//       rounded to bf16 BEFORE the normalisation (what the reference's own bf16 path does: its residual stream is
//       bf16), the normalised value itself is never rounded.
//   residual update               x_out = x_in + alpha * (A . W^T),  xb = bf16(x_out),  stats = row partial sums
//       (blocks.py:146-155) in the epilogue of the out-projection / FFN down-projection / geometric out-projection:
//       no split-K, no partial buffers; per token and 64-channel slice it emits (sum, sum of squares) of the UPDATED
//       row, i.e. the LayerNorm statistics of the next projection, and the bf16 copy that projection multiplies.
//
// A plain block is then 6 launches (projection, qk_prep, attention, residual, projection + SwiGLU, residual) against 11
// with split-K partials, reduce passes and separate add+LayerNorm passes, and x is always current.
#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

constexpr int BK = 64;
constexpr int SM_MAX_PARTS = 64;  // partial sums per row the prologue accepts (d_model <= 4096)
constexpr long long SM_TWO_STAGE_TILES = 280;  // 128 x 128 tiles from which the two-stage, two-workgroups-per-CU variant runs
                                               // (tools/sm_gemm_bench.py with ESMK_SM_TILE = 64 / 128 / 2, cold operands,
                                               // profiles/r03_sm_gemm_variants.txt: behind at 256 tiles, 5-15 % ahead from 288)

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define RAW_BARRIER()                                        \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        __builtin_amdgcn_s_barrier();                        \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)

constexpr int ROLE_RES = 0;  // x_out = x_in + alpha * acc (+ bias), xb, stats
constexpr int ROLE_LN = 1;   // out = epi(rstd * (acc - mean * u) + v)

struct SmArgs {
    const __bf16* A; int lda;       // activations bf16 [M, lda]
    const __bf16* W; int ldw;       // packed weight
    int M, N, ksteps;
    // ROLE_LN
    const float* stats; int parts;  // [M][parts][2] (sum, sum of squares) of the fp32 row
    const float* u; const float* v; // f32 [N] (reference row order); v may be NULL (no beta)
    float eps; int d;               // row width the statistics are over
    void* out;
    // ROLE_RES
    const float* bias; float alpha;
    const float* x_in; float* x_out; __bf16* xb; float* stats_out; int parts_out;
    int n_ctile, n_ttile, nblk;
};

// BW x BT tile, 4 waves (2 x 2): wave tile (BW/2) channels x (BT/2) tokens.
template <int ROLE, int EPI, int BW, int BT, int NS>
__global__ __launch_bounds__(256, (NS * (BW + BT) * 128 > 80 * 1024) ? 1 : 2) void smgemm_kernel(const SmArgs a) {
    constexpr int MI = BW / 32, NJ = BT / 32;
    constexpr int WP = BW / 32, AP = BT / 32;  // LDS-DMA pieces per stage and wave (a piece = 8 rows x 128 B)
    constexpr int P = WP + AP;
    constexpr int SLOT = (BW + BT) * 128;
    static_assert(P * (NS - 1) <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const rowstat = (float*)(smem + NS * SLOT);  // [BT][2] (mean, rstd)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int id = xcd_remap(blockIdx.x, a.nblk);
    // channel-tile-major order: the contiguous id range an XCD walks (xcd_remap) covers FEW channel tiles x ALL token
    // tiles, so every XCD pulls its own slice of the weight from HBM once and keeps it in its L2 while the (small)
    // activation matrix streams by; token-tile-major order made every XCD read the whole weight once per token tile
    const int ctile = id / a.n_ttile, ttile = id - ctile * a.n_ttile;
    const int c0 = ctile * BW, t0 = ttile * BT;
    const int M = a.M, N = a.N, ksteps = a.ksteps;
    const int wr = wave >> 1, wc = wave & 1;
    const int q4 = (lane >> 4) * 4;

    const __bf16* wsrc[WP];
    const __bf16* asrc[AP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (wave * WP + i) * 8 + (lane >> 3);
        wsrc[i] = a.W + (size_t)(c0 + r) * a.ldw + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (wave * AP + i) * 8 + (lane >> 3);
        int tr = t0 + r;
        tr = tr < M ? tr : M - 1;
        asrc[i] = a.A + (size_t)tr * a.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    }
    auto issue = [&](int kt) {
        char* slot = smem + (kt % NS) * SLOT;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i] + kt * BK), LDS_PTR(slot + (wave * WP + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < AP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + kt * BK), LDS_PTR(slot + BW * 128 + (wave * AP + i) * 1024), 16, 0, 0);
    };

    // ROLE_RES: the residual tile this thread updates, requested before anything else
    f32x4 xin[MI][NJ];
    if constexpr (ROLE == ROLE_RES) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int t = t0 + wc * (BT / 2) + j * 16 + (lane & 15);
            t = t < M ? t : M - 1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int c = c0 + wr * (BW / 2) + i * 16 + q4;
                c = c < N ? c : 0;  // N % 4 == 0: c + 3 < N
                xin[i][j] = *(const f32x4*)(a.x_in + (size_t)t * N + c);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < ksteps) issue(i);

    // ROLE_LN: (mean, rstd) of the BT tokens from the producer's partial sums, 256 / BT threads per token; the loads
    // of a thread are all issued before the first is used (one round trip, under the first stages' flight)
    float mean[NJ], rstd[NJ];
    if constexpr (ROLE == ROLE_LN) {
        constexpr int TPT = 256 / BT;  // threads per token
        constexpr int NL = SM_MAX_PARTS / TPT;
        const int r = tid / TPT, sub = tid % TPT;
        int tr = t0 + r;
        tr = tr < M ? tr : M - 1;
        const float* sp = a.stats + (size_t)tr * a.parts * 2;
        f32x2 pv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int p = sub + TPT * i;
            pv[i] = p < a.parts ? *(const f32x2*)(sp + 2 * p) : f32x2{0.f, 0.f};
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            s += pv[i][0];
            q += pv[i][1];
        }
#pragma unroll
        for (int o = 1; o < TPT; o <<= 1) {
            s += __shfl_xor(s, o);
            q += __shfl_xor(q, o);
        }
        if (sub == 0) {
            const float m = s / (float)a.d;
            rowstat[2 * r] = m;
            rowstat[2 * r + 1] = rsqrtf(fmaxf(q / (float)a.d - m * m, 0.f) + a.eps);
        }
        __syncthreads();  // drains the first stages too: once per workgroup
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int tl = wc * (BT / 2) + j * 16 + (lane & 15);
            mean[j] = rowstat[2 * tl];
            rstd[j] = rowstat[2 * tl + 1];
        }
    }

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int aoff[MI], boff[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) aoff[i] = lds_off(wr * (BW / 2) + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = BW * 128 + lds_off(wc * (BT / 2) + j * 16 + (lane & 15), lane >> 4);

    for (int kt = 0; kt < ksteps; ++kt) {
        // stage kt has landed once at most the stages issued behind it are outstanding
        const int after = min(NS - 2, ksteps - 1 - kt);
        if (after >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS >= 5 ? 3 * P : 0) : "memory");
        else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS >= 4 ? 2 * P : 0) : "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RAW_BARRIER();  // every wave's pieces of stage kt landed; everyone has consumed stage kt-1, whose slot is refilled
        if (kt + NS - 1 < ksteps) issue(kt + NS - 1);
        const char* st = smem + (kt % NS) * SLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[MI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + (aoff[i] ^ (kk << 6)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = *(const bf16x8*)(st + (boff[j] ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    if constexpr (ROLE == ROLE_LN) {
        // ---- out = epi(rstd_t * (acc - mean_t * u_n) + v_n) ----
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // channel of acc[i][.][r]: packed row c0 + wr*BW/2 + i*16 + q4 + r; for SwiGLU packed rows alternate in
            // groups of 16 between gate and up (esmk_pack_weight) and u / v are in the reference's row order
            const int pr = c0 + wr * (BW / 2) + i * 16 + q4;
            int cu = pr;
            if constexpr (EPI == ESMK_EPI_SWIGLU) cu = ((pr >> 5) << 4) + (pr & 15) + ((pr & 16) ? (N >> 1) : 0);
            const bool inr = EPI == ESMK_EPI_SWIGLU ? (((pr >> 5) << 4) + (pr & 15) < (N >> 1)) : (pr < N);
            f32x4 uu = f32x4{0.f, 0.f, 0.f, 0.f}, vv = uu;
            if (inr) {
                if (EPI == ESMK_EPI_SWIGLU || (N & 3) == 0) {
                    uu = *(const f32x4*)(a.u + cu);
                    if (a.v) vv = *(const f32x4*)(a.v + cu);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cu + e < N) {
                            uu[e] = a.u[cu + e];
                            if (a.v) vv[e] = a.v[cu + e];
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = rstd[j] * (acc[i][j][e] - mean[j] * uu[e]) + vv[e];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = t0 + wc * (BT / 2) + j * 16 + (lane & 15);
            if (t >= M) continue;
            if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                for (int i = 0; i < MI; i += 2)
                    epi_swiglu4(acc[i][j], acc[i + 1][j], t, ((c0 + wr * (BW / 2)) >> 1) + (i >> 1) * 16 + q4, N >> 1, a.out);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    epi_store4<EPI>(acc[i][j], t, c0 + wr * (BW / 2) + i * 16 + q4, N, nullptr, 1.f, a.out);
            }
        }
    } else {
        // ---- x_out = x_in + alpha * (acc + bias); xb = bf16(x_out); (sum, sum of squares) per 64-channel slice ----
        static_assert(ROLE != ROLE_RES || BW == 64, "one statistics part per 64-channel tile");
        float* const sred = (float*)smem;  // [2 channel halves][BT tokens][2]
        __syncthreads();                   // every wave is done with the ring
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int tl = wc * (BT / 2) + j * 16 + (lane & 15), t = t0 + tl;
            float s = 0.f, q = 0.f;
            if (t < M) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int c = c0 + wr * (BW / 2) + i * 16 + q4;
                    if (c < N) {
                        f32x4 v = acc[i][j];
                        if (a.bias) v += *(const f32x4*)(a.bias + c);
                        f32x4 xv = xin[i][j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            xv[e] += v[e] * a.alpha;
                            s += xv[e];
                            q += xv[e] * xv[e];
                        }
                        *(f32x4*)(a.x_out + (size_t)t * N + c) = xv;
                        if (a.xb) {
                            u32x2 p;
                            p.x = pack_bf16x2(xv[0], xv[1]);
                            p.y = pack_bf16x2(xv[2], xv[3]);
                            *(u32x2*)(a.xb + (size_t)t * N + c) = p;
                        }
                    }
                }
            }
            s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (lane < 16) *(f32x2*)(sred + (wr * BT + tl) * 2) = f32x2{s, q};
        }
        __syncthreads();
        if (a.stats_out && tid < BT && t0 + tid < M) {
            const f32x2 h0 = *(const f32x2*)(sred + tid * 2), h1 = *(const f32x2*)(sred + (BT + tid) * 2);
            *(f32x2*)(a.stats_out + ((size_t)(t0 + tid) * a.parts_out + ctile) * 2) = f32x2{h0[0] + h1[0], h0[1] + h1[1]};
        }
    }
}

// (sum, sum of squares) of every row of x and its bf16 copy: the statistics / operand of a stream that no residual
// epilogue has produced yet (the embedding output in front of block 0).  One wave per row.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int M, int d, float* __restrict__ stats,
                                                        __bf16* __restrict__ xb) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s = 0.f, q = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(x + (size_t)row * d + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += v[e];
            q += v[e] * v[e];
        }
        if (xb) {
            u32x2 p;
            p.x = pack_bf16x2(v[0], v[1]);
            p.y = pack_bf16x2(v[2], v[3]);
            *(u32x2*)(xb + (size_t)row * d + c) = p;
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) *(f32x2*)(stats + (size_t)row * 2) = f32x2{s, q};
}

// W' = W . diag(gamma) rounded to bf16 (one wave per row): u[n] = sum_k bf16(W'[n,k]) - the sum of exactly what the MFMA
// multiplies, so that mean * u cancels the mean's share of the accumulator - and v[n] = sum_k W[n,k] beta[k] in fp32.
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int N, int K,
                                                      float* __restrict__ wf, float* __restrict__ u, float* __restrict__ v) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    float su = 0.f, sv = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float x = w[(size_t)row * K + k];
        const float f = x * gamma[k];
        wf[(size_t)row * K + k] = f;
        su += bf16_bits_to_f32(f32_to_bf16_bits(f));
        if (beta) sv += x * beta[k];
    }
    su = wave_sum(su);
    sv = wave_sum(sv);
    if (lane == 0) {
        u[row] = su;
        if (v) v[row] = sv;
    }
}

template <int ROLE, int EPI, int BW, int BT, int NS>
int launch_sm(SmArgs a, hipStream_t s, const char* what) {
    constexpr int LDS = NS * (BW + BT) * 128 + BT * 8;  // ring | (mean, rstd)
    a.n_ctile = (a.N + BW - 1) / BW;
    a.n_ttile = (a.M + BT - 1) / BT;
    a.nblk = a.n_ctile * a.n_ttile;
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)smgemm_kernel<ROLE, EPI, BW, BT, NS>, LDS, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "%s: cannot reserve %d bytes of LDS", what, LDS);
    hipLaunchKernelGGL((smgemm_kernel<ROLE, EPI, BW, BT, NS>), dim3(a.nblk), dim3(256), LDS, s, a);
    ESMK_CHECK_LAUNCH(what);
    return ESMK_OK;
}

}  // namespace

extern "C" int esmk_res_stats_parts(int N) { return N > 0 ? (N + 63) / 64 : 0; }

extern "C" int esmk_row_stats(const float* x, int M, int d, float* stats, void* xb, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!x || !stats) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_row_stats: null pointer");
    if (M < 0 || d <= 0 || (d & 3) || ((uintptr_t)x & 15) || ((uintptr_t)stats & 7) || ((uintptr_t)xb & 7))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_row_stats: d must be a multiple of 4, x 16-byte, stats and xb 8-byte aligned");
    hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, M, d, stats, (__bf16*)xb);
    ESMK_CHECK_LAUNCH("esmk_row_stats");
    return ESMK_OK;
}

extern "C" int esmk_fold_layernorm(const float* w, const float* gamma, const float* beta, int N, int K, float* w_folded,
                                   float* u, float* v, esmk_stream_t stream) {
    if (!w || !gamma || !w_folded || !u || (beta && !v)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_fold_layernorm: null pointer");
    if (N <= 0 || K <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_fold_layernorm: bad shape");
    hipLaunchKernelGGL(fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, N, K, w_folded, u,
                       beta ? v : nullptr);
    ESMK_CHECK_LAUNCH("esmk_fold_layernorm");
    return ESMK_OK;
}

extern "C" int esmk_gemm_lnfold(const void* xb, int lda, const float* stats, int parts, float eps, const void* W_folded,
                                const float* u, const float* v, int M, int N, int d, int epilogue, void* out,
                                esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!xb || !stats || !W_folded || !u || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_lnfold: null pointer");
    if (M < 0 || N <= 0 || d <= 0 || (d % BK) || parts < 1 || parts > SM_MAX_PARTS)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_lnfold: bad shape M=%d N=%d d=%d parts=%d (d a multiple of 64, 1 <= parts <= %d)",
                  M, N, d, parts, SM_MAX_PARTS);
    if (lda < d || (lda & 7)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_lnfold: lda=%d must be >= d and a multiple of 8", lda);
    if (((uintptr_t)u & 15) || ((uintptr_t)v & 15) || ((uintptr_t)stats & 7))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_lnfold: u / v must be 16-byte aligned, stats 8-byte aligned");
    SmArgs a{};
    a.A = (const __bf16*)xb; a.lda = lda; a.W = (const __bf16*)W_folded; a.ldw = d; a.M = M; a.N = N; a.ksteps = d / BK;
    a.stats = stats; a.parts = parts; a.u = u; a.v = v; a.eps = eps; a.d = d; a.out = out;
    hipStream_t s = (hipStream_t)stream;
    // tile: 128 x 64 tokens, three stages (two workgroups per CU) while that is what it takes to give every CU work;
    // 128 x 128, three stages, one workgroup per CU (half the staged bytes per FLOP) from 224 tiles; 128 x 128 with TWO
    // stages, two workgroups per CU, once there are about two such tiles per CU (mid-sized batches: the partner workgroup
    // covers a stage's round trip better than a third stage does)
    const int sw = esmk_switch("ESMK_SM_TILE", 0);  // experiments: 64 / 128 / 2 force a variant
    const long long tiles128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
    const int variant = sw ? sw : tiles128 >= SM_TWO_STAGE_TILES ? 2 : tiles128 >= 224 ? 128 : 64;
#define LNFOLD_CASE(EPI)                                                                                       \
    return variant == 2     ? launch_sm<ROLE_LN, EPI, 128, 128, 2>(a, s, "esmk_gemm_lnfold")                   \
           : variant == 128 ? launch_sm<ROLE_LN, EPI, 128, 128, 3>(a, s, "esmk_gemm_lnfold")                   \
                            : launch_sm<ROLE_LN, EPI, 128, 64, 3>(a, s, "esmk_gemm_lnfold")
    switch (epilogue) {
        case ESMK_EPI_BF16: LNFOLD_CASE(ESMK_EPI_BF16);
        case ESMK_EPI_F32: LNFOLD_CASE(ESMK_EPI_F32);
        case ESMK_EPI_SWIGLU:
            if (N % 32) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm_lnfold: SwiGLU epilogue needs N %% 32 == 0");
            LNFOLD_CASE(ESMK_EPI_SWIGLU);
        default: ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm_lnfold: epilogue %d is not available with the folded LayerNorm", epilogue);
    }
#undef LNFOLD_CASE
}

extern "C" int esmk_gemm_residual_stats(const void* A, int lda, const void* W, int M, int N, int K, const float* bias,
                                        float alpha, const float* x_in, float* x_out, void* xb, float* stats,
                                        esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!A || !W || !x_in || !x_out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_residual_stats: null pointer");
    if (M < 0 || N <= 0 || K <= 0 || (N & 3)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_residual_stats: bad shape M=%d N=%d K=%d (N %% 4 == 0)", M, N, K);
    const int Kpad = round_up(K, BK);
    if (lda < Kpad || (lda & 7))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_residual_stats: lda=%d must be >= round_up(K,64)=%d and a multiple of 8", lda, Kpad);
    if (((uintptr_t)x_in & 15) || ((uintptr_t)x_out & 15) || ((uintptr_t)xb & 7) || ((uintptr_t)stats & 7) || ((uintptr_t)bias & 15))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_residual_stats: x / bias must be 16-byte aligned, xb / stats 8-byte aligned");
    SmArgs a{};
    a.A = (const __bf16*)A; a.lda = lda; a.W = (const __bf16*)W; a.ldw = Kpad; a.M = M; a.N = N; a.ksteps = Kpad / BK;
    a.bias = bias; a.alpha = alpha; a.x_in = x_in; a.x_out = x_out; a.xb = (__bf16*)xb; a.stats_out = stats;
    a.parts_out = esmk_res_stats_parts(N);
    return launch_sm<ROLE_RES, ESMK_EPI_RESIDUAL, 64, 64, 4>(a, (hipStream_t)stream, "esmk_gemm_residual_stats");
}
