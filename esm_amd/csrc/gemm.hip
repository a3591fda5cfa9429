// bf16 MFMA GEMM with fused epilogues:  out = epi(A[M,K] . W[N,K]^T)
//
// Orientation: the MFMA "M" dimension is the OUTPUT CHANNEL (rows of W) and the MFMA "N" dimension is the
// TOKEN, i.e. each workgroup computes a (channels x tokens) tile of out^T.  With v_mfma_f32_16x16x32_bf16
// a lane then holds 4 consecutive output channels of ONE token in one accumulator, so every epilogue
// (bf16 store, SwiGLU gate*up, fp32 residual read-modify-write) works on 8/16-byte row-major pieces.
//
// Tile: 128 channels x 128 tokens x BK=64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
// Both operands are K-contiguous in memory, staged global->LDS with global_load_lds_dwordx4 (no VGPR
// round trip) into double-buffered 16 KiB tiles; one barrier per K-step.  LDS rows are 128 B; the 16-byte
// chunk index is XOR-swizzled with (row>>1)&7 - applied on the per-lane SOURCE address (the DMA writes
// LDS linearly) and on the ds_read_b128 address - which makes the 16-lane ds_read_b128 groups
// conflict-free.
#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

constexpr int BW = 128;  // output channels per block
// tokens per block: template parameter BT = 128 (default) or 64 (remainder columns of the split dispatch, where
// 128-token tiles would leave one 4-wave workgroup per CU and nothing to overlap its barrier stalls with)
constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}


constexpr int EPI_PARTIAL = 100;  // split-K: fp32 partial sums [split][M][n_ctile*128], no epilogue

// NS = LDS stages: 2 (two workgroups per CU overlap each other's barrier stalls) or 4 (one workgroup per CU with
// three K-steps of loads in flight: small grids, where every K-step otherwise pays a full memory latency)
template <int EPI, int BT, int NS = 2>
__global__ __launch_bounds__(256, (NS * (TILE_BYTES + BT * 128) <= 80 * 1024) ? 2 : 1) void gemm_kernel(const __bf16* __restrict__ A, int lda,
                                                      const __bf16* __restrict__ W, int ldw, int M, int N,
                                                      int ksteps, const float* __restrict__ bias, float alpha,
                                                      void* __restrict__ out, int n_ctile, int nblk, int ct0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][W tile | A tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // split-K: blockIdx.y = split index; this workgroup reduces K-steps [kt0, kt1) of its tile
    const int kt0 = (EPI == EPI_PARTIAL) ? (int)((long long)ksteps * blockIdx.y / gridDim.y) : 0;
    const int kt1 = (EPI == EPI_PARTIAL) ? (int)((long long)ksteps * (blockIdx.y + 1) / gridDim.y) : ksteps;
    const int id = xcd_remap(blockIdx.x, nblk);
    const int ttile = id / n_ctile, ctile = id - ttile * n_ctile;
    const int c0 = (ct0 + ctile) * BW, t0 = ttile * BT;  // ct0: first channel tile of this launch (split dispatch)

    // ---- staging addresses: wave w owns pieces 4w..4w+3 (8 rows x 128 B each) of both tiles ----
    constexpr int AP = BT / 32;  // A pieces per wave (a piece = 8 rows x 128 B)
    constexpr int NJ = BT / 32;  // 16-token MFMA columns per wave
    const __bf16* wsrc[4];
    const __bf16* asrc[AP];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        wsrc[i] = W + (size_t)(c0 + r) * ldw + lc * 8;
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int r = (wave * AP + i) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((r >> 1) & 7);
        int tr = t0 + r;
        tr = tr < M ? tr : M - 1;
        asrc[i] = A + (size_t)tr * lda + lc * 8;
    }
    constexpr int STAGE_BYTES = TILE_BYTES + BT * 128;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[i] + kt * BK), LDS_PTR(base + wave * 4096 + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < AP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + kt * BK),
                                             LDS_PTR(base + TILE_BYTES + (wave * AP + i) * 1024), 16, 0, 0);
    };

    const int wr = wave >> 1, wc = wave & 1;  // wave's sub-tile: channels wr*64..+63, tokens wc*(BT/2)..
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) inside a tile for kk = 0; kk = 1 flips chunk bit 2
    int aoff[4], boff[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = lds_off(wr * 64 + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = lds_off(wc * (BT / 2) + j * 16 + (lane & 15), lane >> 4);

    constexpr int LPS = 4 + AP;  // global_load_lds per wave and stage
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (kt0 + i < kt1) stage(i, kt0 + i);
    for (int kt = kt0; kt < kt1; ++kt) {
        // wait for this wave's loads of stage kt: all but the stages issued after it may stay in flight
        if constexpr (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int after = min(NS - 2, kt1 - 1 - kt);
            if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // stage kt landed for every wave; everyone is done reading the buffer refilled next.  A RAW barrier: the fence
        // inside __syncthreads() is s_waitcnt vmcnt(0), which would drain the stages in flight (NS > 2) at every step
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NS - 1 < kt1) stage((kt + NS - 1 - kt0) % NS, kt + NS - 1);
        const char* wt = smem + ((kt - kt0) % NS) * STAGE_BYTES;
        const char* at = wt + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(wt + (aoff[i] ^ (kk << 6)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = *(const bf16x8*)(at + (boff[j] ^ (kk << 6)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc[i][j][r] = out^T[channel = c0 + wr*64 + i*16 + (lane>>4)*4 + r][token = ... + (lane&15)] ----
    const int q4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = t0 + wc * (BT / 2) + j * 16 + (lane & 15);
        if (t >= M) continue;
        if constexpr (EPI == EPI_PARTIAL) {
            const int ldp = n_ctile * BW;  // ct0 == 0 here
            float* o = (float*)out + ((size_t)blockIdx.y * M + t) * ldp + c0 + wr * 64 + q4;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(o + i * 16) = acc[i][j];
        } else if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
            for (int i = 0; i < 4; i += 2)
                epi_swiglu4(acc[i][j], acc[i + 1][j], t, ((c0 + wr * 64) >> 1) + (i >> 1) * 16 + q4, N >> 1, out, bias);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) epi_store4<EPI>(acc[i][j], t, c0 + wr * 64 + i * 16 + q4, N, bias, alpha, out);
        }
    }
}

// fp32 [N,K] row-major -> bf16 [Npad, Kpad], optional SwiGLU row interleave (groups of 16 gate rows then the
// 16 matching up rows).  One thread per 8 output elements.
__global__ void pack_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int N, int K, int Npad,
                                   int Kpad, int mode) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kchunks = Kpad >> 3;
    const size_t total = (size_t)Npad * kchunks;
    if (idx >= total) return;
    const int prow = (int)(idx / kchunks);
    const int k0 = (int)(idx - (size_t)prow * kchunks) * 8;
    int srow = prow;
    if (mode == ESMK_PACK_SWIGLU) {
        const int h = N >> 1;
        const int g = prow >> 5, j = prow & 31;
        srow = (j < 16) ? (g * 16 + j) : (h + g * 16 + (j - 16));
        if (g * 16 + (j & 15) >= h) srow = N;  // beyond -> zero
    }
    unsigned int pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = 0.f, b = 0.f;
        if (srow < N) {
            if (k0 + 2 * e < K) a = w[(size_t)srow * K + k0 + 2 * e];
            if (k0 + 2 * e + 1 < K) b = w[(size_t)srow * K + k0 + 2 * e + 1];
        }
        pk[e] = pack_bf16x2(a, b);
    }
    *(u32x4*)(out + (size_t)prow * Kpad + k0) = u32x4{pk[0], pk[1], pk[2], pk[3]};
}

// Covers channel tiles [ct0, round_up(N,128)/128) of the output.
template <int EPI, int BT = 128, int NS = 2>
int launch_gemm_ns(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, float alpha, void* out,
                   hipStream_t s, int ct0) {
    const int Kpad = round_up(K, BK);
    const int n_ctile = round_up(N, BW) / BW - ct0;
    const int n_ttile = (M + BT - 1) / BT;
    const int nblk = n_ctile * n_ttile;
    constexpr int LDS = NS * (TILE_BYTES + BT * 128);
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm_kernel<EPI, BT, NS>, LDS, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", LDS);
    hipLaunchKernelGGL((gemm_kernel<EPI, BT, NS>), dim3(nblk), dim3(256), LDS, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, bias, alpha, out, n_ctile, nblk, ct0);
    ESMK_CHECK_LAUNCH("esmk_gemm");
    return ESMK_OK;
}

template <int EPI, int BT = 128>
int launch_gemm(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, float alpha, void* out,
                hipStream_t s, int ct0 = 0) {
    if (esmk_switch("ESMK_GEMM_NS", 2) == 3) return launch_gemm_ns<EPI, BT, 3>(A, lda, W, M, N, K, bias, alpha, out, s, ct0);
    const int Kpad = round_up(K, BK);
    const int n_ctile = round_up(N, BW) / BW - ct0;
    const int n_ttile = (M + BT - 1) / BT;
    const int nblk = n_ctile * n_ttile;
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm_kernel<EPI, BT>, 2 * (TILE_BYTES + BT * 128), lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", 2 * (TILE_BYTES + BT * 128));
    hipLaunchKernelGGL((gemm_kernel<EPI, BT>), dim3(nblk), dim3(256), 2 * (TILE_BYTES + BT * 128), s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, bias, alpha, out, n_ctile, nblk, ct0);
    ESMK_CHECK_LAUNCH("esmk_gemm");
    return ESMK_OK;
}

// Split-K second pass: sum the fp32 partials of every split in a fixed order (deterministic) and apply the
// epilogue.  One thread per 4 consecutive output channels of one token.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N, int ldp,
                                                            const float* __restrict__ bias, float alpha,
                                                            void* __restrict__ out) {
    const int ow4 = ((EPI == ESMK_EPI_SWIGLU ? (N >> 1) : N) + 3) >> 2;  // groups of 4 output channels per token
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * ow4) return;
    const int t = (int)(idx / ow4), c = (int)(idx - (size_t)t * ow4) * 4;
    const size_t sstride = (size_t)M * ldp;
    if constexpr (EPI == ESMK_EPI_SWIGLU) {
        // output channel c -> packed gate row (c/16)*32 + c%16, up row = gate row + 16 (esmk_pack_weight)
        const float* p = part + (size_t)t * ldp + (c >> 4) * 32 + (c & 15);
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f}, u = g;
        for (int s = 0; s < S; ++s) {
            g += *(const f32x4*)(p + s * sstride);
            u += *(const f32x4*)(p + s * sstride + 16);
        }
        epi_swiglu4(g, u, t, c, N >> 1, out, bias);
    } else {
        const float* p = part + (size_t)t * ldp + c;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) v += *(const f32x4*)(p + s * sstride);
        epi_store4<EPI>(v, t, c, N, bias, alpha, out);
    }
}

// How many K splits a small problem gets: enough workgroups for ~1.5 per CU, at least 4 K-steps each.
int splitk_factor(int M, int N, int K) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int ksteps = round_up(K, BK) / BK;
    if (tiles >= 192 || ksteps < 8) return 1;
    int s = 384 / tiles;
    if (s > ksteps / 4) s = ksteps / 4;
    if (s > 8) s = 8;
    return s < 2 ? 1 : s;
}

template <int EPI>
int launch_splitk(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, float alpha, void* out,
                  hipStream_t s, int S, float* part) {
    const int Kpad = round_up(K, BK);
    const int n_ctile = round_up(N, BW) / BW;
    const int n_ttile = (M + 127) / 128;
    const int nblk = n_ctile * n_ttile;
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm_kernel<EPI_PARTIAL, 128>, 4 * TILE_BYTES, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", 4 * TILE_BYTES);
    // two stages, two workgroups per CU: measured faster than four stages at one workgroup per CU (ESM3 1 x 512:
    // 6.9 vs 8.0 ms per forward) - with one wave per SIMD nothing overlaps the LDS reads with the MFMAs
    hipLaunchKernelGGL((gemm_kernel<EPI_PARTIAL, 128>), dim3(nblk, S), dim3(256), 4 * TILE_BYTES, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, nullptr, 1.f, (void*)part, n_ctile, nblk, 0);
    ESMK_CHECK_LAUNCH("esmk_gemm(split-K)");
    const int ow4 = ((EPI == ESMK_EPI_SWIGLU ? (N >> 1) : N) + 3) >> 2;
    const size_t total = (size_t)M * ow4;
    hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, S, M, N,
                       n_ctile * BW, bias, alpha, out);
    ESMK_CHECK_LAUNCH("esmk_gemm(split-K reduce)");
    return ESMK_OK;
}

}  // namespace

int esmk_gemm256(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                 float alpha, void* out, hipStream_t s, int n_ctile);
int esmk_gemm288(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s);
int esmk_gemm192(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s);
int esmk_gemm288_stats(const void* A, int lda, const void* W, int M, int d, int K, void* out, float* stats, int tc,
                       hipStream_t s);
int esmk_gemm256_stats(const void* A, int lda, const void* W, int M, int d, void* out, float* stats, hipStream_t s,
                       int n_ctile);

#ifdef ESMK_EXPERIMENTS
// Alternative kernels of the experiments build only (libesmk_x.so): the four-wave 128x128-wave-tile kernels
// (gemm4w.hip: LDS-DMA operands, ESMK_GEMM_4W=1; gemm4v.hip: operands staged through VGPRs, ESMK_GEMM_4W=2) and the
// persistent cross-tile-prefetch kernel (gemm256p.hip, ESMK_GEMM_PERSIST=1).  All measured slower than the product
// kernels; kept for A/B runs and covered by tests/test_gpu_kernels.py through libesmk_x.so.
int esmk_gemm4w(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                float alpha, void* out, hipStream_t s, int n_ctile);
int esmk_gemm4v(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                float alpha, void* out, hipStream_t s, int n_ctile);
int esmk_gemm256p(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, float alpha, void* out,
                  hipStream_t s);
#endif

static int gemm256_tile(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                        float alpha, void* out, hipStream_t s, int n_ctile) {
#ifdef ESMK_EXPERIMENTS
    const int v = esmk_switch("ESMK_GEMM_4W", 0);
    if (v == 2) return esmk_gemm4v(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, n_ctile);
    if (v) return esmk_gemm4w(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, n_ctile);
#endif
    return esmk_gemm256(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, n_ctile);
}

extern "C" size_t esmk_packed_weight_bytes(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)round_up(N, 256) * round_up(K, BK) * 2;
}

extern "C" int esmk_pack_weight(const float* w, void* packed, int N, int K, int mode, esmk_stream_t stream) {
    if (!w || !packed || N <= 0 || K <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pack_weight: null pointer or empty shape");
    if (mode != ESMK_PACK_PLAIN && mode != ESMK_PACK_SWIGLU) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pack_weight: bad mode");
    if (mode == ESMK_PACK_SWIGLU && (N % 32) != 0)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_pack_weight: SwiGLU packing needs N %% 32 == 0 (got %d)", N);
    const int Npad = round_up(N, 256), Kpad = round_up(K, BK);
    const size_t total = (size_t)Npad * (Kpad / 8);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16*)packed, N, K, Npad, Kpad, mode);
    ESMK_CHECK_LAUNCH("esmk_pack_weight");
    return ESMK_OK;
}

extern "C" size_t esmk_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int S = splitk_factor(M, N, K);
    return S > 1 ? (size_t)S * M * round_up(N, BW) * sizeof(float) : 0;
}

extern "C" int esmk_gemm(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                         float alpha, void* out, esmk_stream_t stream) {
    return esmk_gemm_ws(A, lda, W, M, N, K, epilogue, bias, alpha, out, nullptr, 0, stream);
}

// ---- allocation-free form (esmk_gemm_noalloc) -------------------------------------------------------------------
// The split-K partials live in a device array that belongs to the library instead of a caller's workspace, for
// callers that cannot allocate (HIP graph capture, serving loops).  `scratch` below is only the HOST-side handle of
// that array: its device address differs per GPU and comes from hipGetSymbolAddress, once per device.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)ESMK_GEMM_NOALLOC_MAX_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static void* g_scratch_ptr[64];                                // device address of `scratch` on device i
static std::atomic<unsigned long long> g_scratch_known{0};     // bit i: g_scratch_ptr[i] is valid

static int noalloc_scratch_ptr(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm_noalloc: cannot identify the current device (index %d)", dev);
    const unsigned long long bit = 1ull << dev;
    if (!(g_scratch_known.load(std::memory_order_acquire) & bit)) {
        void* p = nullptr;  // a lost race between two threads stores the same address twice
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
        g_scratch_known.fetch_or(bit, std::memory_order_release);
    }
    *out = g_scratch_ptr[dev];
    return ESMK_OK;
}

extern "C" int esmk_gemm_noalloc(const void* A, int lda, const void* W, int M, int N, int K, int epilogue,
                                 const float* bias, float alpha, void* out, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (M < 0 || N <= 0 || K <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_noalloc: bad shape M=%d N=%d K=%d", M, N, K);
    const size_t need = esmk_gemm_workspace_bytes(M, N, K);  // 0: the shape is not split, no scratch is touched
    if (need > NOALLOC_SCRATCH_BYTES)
        ESMK_FAIL(ESMK_ERR_WORKSPACE, "esmk_gemm_noalloc: M=%d N=%d K=%d needs %zu bytes of split-K partials, the library's "
                  "fixed array holds %zu - use esmk_gemm_ws with a workspace of your own", M, N, K, need, NOALLOC_SCRATCH_BYTES);
    void* part = nullptr;
    if (need) {
        const int rc = noalloc_scratch_ptr(&part);
        if (rc != ESMK_OK) return rc;
    }
    // every partial the reduce pass reads is written by the first pass of the same call: nothing needs clearing
    return esmk_gemm_ws(A, lda, W, M, N, K, epilogue, bias, alpha, out, part, need, stream);
}

// Channel-tile width for a LARGE bf16-output problem without bias: 288, 192 or 256; 0 = not large (the 128-tile kernels
// and split-K below take it).  Cost model: workgroups run one per CU, so a launch takes ceil(tiles / 256) rounds of a
// duration proportional to the tile width; measured on 19 shapes with N % 576 == 0 (tools/gemm_tile_pick.py,
// profiles/r03_gemm_tile_pick.txt) the model orders the three widths correctly wherever they differ by more than 3 %
// up to 4 rounds; beyond that the rounds blur and the widest tile without a ragged column (288) stays best.
//   N % 288 == 0 (and not % 256): 288, unless at <= 4 rounds a narrower width needs less (8192 x 1152: 192-wide is one
//     full round, 288-wide one half-empty one: 32.0 -> 25.3 us; 4096 x 3456: 256-wide 34.6 -> 32.8 us)
//   else N % 192 == 0 and 192-wide needs less than 256-wide (d_model 1536 at 8192 tokens): 192
//   else 256 when there are at least 128 tiles.
static int wide_tile_pick(int M, int N) {
    if (!esmk_switch("ESMK_GEMM_288", 1)) return 0;
    const long long nt = (M + 255) / 256, nc256 = (N + 255) / 256;
    auto rounds = [&](long long cols) { return (nt * cols + 255) / 256; };
    const bool big = nt * nc256 >= 128;
    const bool ok288 = (N % 288) == 0 && (N % 256) != 0 && nt * (N / 288) >= 128;
    const bool ok192 = (N % 192) == 0 && nt * (N / 192) >= 128;
    const long long c256 = rounds(nc256) * 256, c192 = ok192 ? rounds(N / 192) * 192 : 0;
    if (esmk_switch("ESMK_GEMM_192", 0) && ok192) return 192;
    if (ok288) {
        const long long r288 = rounds(N / 288);
        long long best = r288 * 288;
        int pick = 288;
        if (r288 <= 4 && esmk_switch("ESMK_GEMM_PICK", 1)) {
            if (big && c256 < best) best = c256, pick = 256;   // on a tie with 192-wide tiles the wider ones are not slower
            if (ok192 && c192 < best) best = c192, pick = 192;
        }
        return pick;
    }
    if (ok192 && c192 < c256) return 192;
    return big ? 256 : 0;
}

// Channel-tile width of the kernel that computes the QKV projection (N = 3 d, K = d, bf16, no bias) AND emits the q-side
// LayerNorm partial sums; 0 when no such kernel takes the shape.  288 / 192: when esmk_gemm's dispatch picks that width
// and its tiles do not straddle the q / k boundary (one part per wave half-tile).  256: the 256-tile kernel with 64- or
// 128-channel parts, where the dispatch picks 256-wide tiles, or 192-wide ones that would straddle (d_model 2560).
// Where it picks 288-wide tiles that straddle (d_model 960 at >= 12288 tokens) the three-call path stays: measured,
// the narrower tiles cost more than the fused q side saves (tools/qfuse_ab.py: 279.8 vs 284.4 us at 64 x 512).
static int qkv_stats_tile(int M, int d) {
    if (esmk_switch("ESMK_GEMM_TILE", 0) != 0) return 0;
    const int pick = wide_tile_pick(M, 3 * d);
    if (pick == 288) return (d % 288) == 0 ? 288 : 0;
    if (pick == 192 && (d % 192) == 0) return 192;
    const int stats256 = esmk_switch("ESMK_QKV_STATS256", 1);
    if (pick && stats256 != 0 && (d % 64) == 0 && (d <= 2048 || ((d % 128) == 0 && d <= 4096)) &&
        (long long)((M + 255) / 256) * ((3 * d + 255) / 256) >= 128)
        return 256;
    return 0;
}

static int qkv_stats_parts_of(int tc, int d) { return tc == 256 ? (d > 2048 ? d / 128 : d / 64) : tc ? 2 * (d / tc) : 0; }

extern "C" int esmk_qkv_stats_parts(int M, int d_model) {
    if (M <= 0 || d_model <= 0) return 0;
    return qkv_stats_parts_of(qkv_stats_tile(M, d_model), d_model);
}

extern "C" int esmk_gemm_qkv_stats(const void* A, int lda, const void* W, int M, int d_model, void* out, float* qstats,
                                   esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!A || !W || !out || !qstats) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_qkv_stats: null pointer");
    if (M < 0 || d_model <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_qkv_stats: bad shape M=%d d=%d", M, d_model);
    if (lda < round_up(d_model, BK) || (lda & 7))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm_qkv_stats: lda=%d must be >= round_up(d,64) and a multiple of 8", lda);
    const int tc = qkv_stats_tile(M, d_model);
    if (!tc)
        ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm_qkv_stats: shape M=%d d=%d has no statistics-emitting kernel "
                  "(esmk_qkv_stats_parts() == 0): use esmk_gemm + esmk_qk_prep", M, d_model);
    if (tc != 256) return esmk_gemm288_stats(A, lda, W, M, d_model, d_model, out, qstats, tc, (hipStream_t)stream);
    // 256-wide tiles, with esmk_gemm's split dispatch: a last column that is at most half full goes to the 128-tile
    // kernel when that saves a round of workgroups (it holds v columns only: (nc - 1) * 256 >= 3 d - 128 >= 2 d)
    const int N = 3 * d_model, nt = (M + 255) / 256, nc = (N + 255) / 256, rem = N - (nc - 1) * 256;
    const bool split = esmk_switch("ESMK_GEMM_SPLIT", 1) && nc >= 2 && rem <= 128 && (nc - 1) * 256 >= 2 * d_model &&
                       (nt * (nc - 1) + 255) / 256 < (nt * nc + 255) / 256;
    const int rc = esmk_gemm256_stats(A, lda, W, M, d_model, out, qstats, (hipStream_t)stream, split ? nc - 1 : nc);
    if (rc != ESMK_OK || !split) return rc;
    return launch_gemm<ESMK_EPI_BF16, 64>(A, lda, W, M, N, d_model, nullptr, 1.f, out, (hipStream_t)stream, (nc - 1) * 2);
}

extern "C" int esmk_gemm_ws(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                            float alpha, void* out, void* workspace, size_t workspace_bytes, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!A || !W || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: null pointer");
    if (M < 0 || N <= 0 || K <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    if (M == 0) return ESMK_OK;
    if (lda < round_up(K, BK) || (lda & 7))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: lda=%d must be >= round_up(K,64)=%d and a multiple of 8", lda,
                  round_up(K, BK));
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == ESMK_EPI_SWIGLU && (N % 32)) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm: SwiGLU epilogue needs N %% 32 == 0");
    if (epilogue == ESMK_EPI_RESIDUAL && (N % 4)) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm: residual epilogue needs N %% 4 == 0");
    if (epilogue == ESMK_EPI_BIAS_GELU && !bias) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: bias+gelu epilogue needs a bias");
    {
        // large problems: 256x256 ping-pong kernel; small / skinny ones: 128x128 (more, smaller workgroups)
        const int ov = esmk_switch("ESMK_GEMM_TILE", 0);  // experiments: 128 / 256 force a tile family
        // measured crossover (tools/gemm_bench.py, M = 512 ... 32768): the 256-tile kernel wins once there are
        // about half as many 256x256 tiles as CUs; below that the 128-tile kernel fills the chip better.
        // a bias with the bf16 epilogue (bias=True block variants) is applied by the 128-tile kernel and the split-K
        // reduce only (the 256-tile kernel has biased instantiations of its other epilogues)
        const bool bias128 = bias && epilogue == ESMK_EPI_BF16;
        const bool big = !bias128 && (long long)((M + 255) / 256) * ((N + 255) / 256) >= 128;
        // bf16 output without bias: 288- / 192- / 256-wide tiles by the round model of wide_tile_pick()
        int pick = 0;
        if (ov == 0 && !bias && epilogue == ESMK_EPI_BF16) {
            pick = wide_tile_pick(M, N);
            if (pick == 288) return esmk_gemm288(A, lda, W, M, N, K, alpha, out, s);
            if (pick == 192) return esmk_gemm192(A, lda, W, M, N, K, alpha, out, s);
        }
        if ((ov == 256 && !bias128) || (ov == 0 && (big || pick == 256))) {
            const int persist = esmk_switch("ESMK_GEMM_PERSIST", 0);
#ifdef ESMK_EXPERIMENTS
            if (persist && (epilogue == ESMK_EPI_BF16 || epilogue == ESMK_EPI_SWIGLU)) {
                const int rc = esmk_gemm256p(A, lda, W, M, N, K, epilogue, alpha, out, s);
                if (rc <= 0) return rc;  // 1 = shape not eligible for the persistent kernel
            }
#endif
            // Split dispatch: when the last 256-channel tile column is at most half full (N = 1152: 4.5 columns)
            // and dropping it saves a whole round of workgroups over the 256 CUs (640 -> 512 tiles: 3 -> 2
            // rounds), the 256-kernel covers the full columns and the 128-tile kernel the remaining <= 128
            // channels; its small workgroups start as soon as CUs drain from the last full round.
            const int nt = (M + 255) / 256, nc = (N + 255) / 256;
            const int rem = N - (nc - 1) * 256;
            const int split_on = esmk_switch("ESMK_GEMM_SPLIT", 1);
            const bool split = split_on && !persist && nc >= 2 && rem <= 128 && epilogue != ESMK_EPI_SWIGLU &&
                               (nt * (nc - 1) + 255) / 256 < (nt * nc + 255) / 256;
            if (!split) return gemm256_tile(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, nc);
            const int rc = gemm256_tile(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, nc - 1);
            if (rc != ESMK_OK) return rc;
            const int ct0 = (nc - 1) * 2;
            switch (epilogue) {
                case ESMK_EPI_BF16: return launch_gemm<ESMK_EPI_BF16, 64>(A, lda, W, M, N, K, bias, alpha, out, s, ct0);
                case ESMK_EPI_F32: return launch_gemm<ESMK_EPI_F32, 64>(A, lda, W, M, N, K, bias, alpha, out, s, ct0);
                case ESMK_EPI_RESIDUAL: return launch_gemm<ESMK_EPI_RESIDUAL, 64>(A, lda, W, M, N, K, bias, alpha, out, s, ct0);
                case ESMK_EPI_BIAS_GELU: return launch_gemm<ESMK_EPI_BIAS_GELU, 64>(A, lda, W, M, N, K, bias, alpha, out, s, ct0);
                default: ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: unknown epilogue %d", epilogue);
            }
        }
    }
    // small problems with a short K (<= 20 K-steps: d_model <= 1280 as K) and at least 128 tiles of 128 x 64: those tiles
    // (twice the workgroups, no partials, no reduce launch) instead of split-K.  Measured with operands coming from HBM,
    // as inside a forward (tools/small_gemm_ab.py): 2048 x 960 x 960 24.5 -> 18.2 us, 1024 x 2880 x 960 28.2 -> 20.6,
    // 2048 x 1152 x 1152 26.0 -> 21.7; with fewer tiles (512 x 1536 x 1536: 21.9 vs 25.5) or a longer K (512 x 1536 x
    // 4096: 22.4 vs 38.3) too few workgroups pull the weights and split-K below stays ahead.  C1 (8 x 256, d 960) forward:
    // 3.79 -> 3.67 ms in one process.
    {
        const long long tiles128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
        const long long tiles64 = (long long)((M + 63) / 64) * ((N + 127) / 128);
        const int ksteps = round_up(K, BK) / BK;
        if (esmk_switch("ESMK_GEMM_TILE", 0) == 0 && esmk_switch("ESMK_GEMM_BT64", 1) && tiles128 < 192 && tiles64 >= 128 &&
            ksteps <= 20) {
            switch (epilogue) {
                case ESMK_EPI_BF16: return launch_gemm<ESMK_EPI_BF16, 64>(A, lda, W, M, N, K, bias, alpha, out, s);
                case ESMK_EPI_F32: return launch_gemm<ESMK_EPI_F32, 64>(A, lda, W, M, N, K, bias, alpha, out, s);
                case ESMK_EPI_RESIDUAL: return launch_gemm<ESMK_EPI_RESIDUAL, 64>(A, lda, W, M, N, K, bias, alpha, out, s);
                case ESMK_EPI_BIAS_GELU: return launch_gemm<ESMK_EPI_BIAS_GELU, 64>(A, lda, W, M, N, K, bias, alpha, out, s);
                default: break;  // SwiGLU: measured slower on 64-token tiles (512 x 8192 x 1536: 25.4 -> 31.5 us)
            }
        }
    }
    // small problems: split K over several workgroups per tile when the caller lent a workspace for the partials
    // (not for the read-modify-write residual epilogue: its output is also an input)
    const int splitk_on = esmk_switch("ESMK_GEMM_SPLITK", 1);
    const int S = (splitk_on && workspace && epilogue != ESMK_EPI_RESIDUAL && esmk_switch("ESMK_GEMM_TILE", 0) == 0) ? splitk_factor(M, N, K) : 1;
    if (S > 1 && workspace_bytes >= esmk_gemm_workspace_bytes(M, N, K) && !((uintptr_t)workspace & 15)) {
        float* part = (float*)workspace;
        switch (epilogue) {
            case ESMK_EPI_BF16: return launch_splitk<ESMK_EPI_BF16>(A, lda, W, M, N, K, bias, alpha, out, s, S, part);
            case ESMK_EPI_F32: return launch_splitk<ESMK_EPI_F32>(A, lda, W, M, N, K, bias, alpha, out, s, S, part);
            case ESMK_EPI_SWIGLU: return launch_splitk<ESMK_EPI_SWIGLU>(A, lda, W, M, N, K, bias, alpha, out, s, S, part);
            case ESMK_EPI_BIAS_GELU: return launch_splitk<ESMK_EPI_BIAS_GELU>(A, lda, W, M, N, K, bias, alpha, out, s, S, part);
            default: ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: unknown epilogue %d", epilogue);
        }
    }
    switch (epilogue) {
        case ESMK_EPI_BF16: return launch_gemm<ESMK_EPI_BF16>(A, lda, W, M, N, K, bias, alpha, out, s);
        case ESMK_EPI_F32: return launch_gemm<ESMK_EPI_F32>(A, lda, W, M, N, K, bias, alpha, out, s);
        case ESMK_EPI_SWIGLU:
            if (N % 32) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm: SwiGLU epilogue needs N %% 32 == 0");
            return launch_gemm<ESMK_EPI_SWIGLU>(A, lda, W, M, N, K, bias, alpha, out, s);
        case ESMK_EPI_RESIDUAL:
            if (N % 4) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_gemm: residual epilogue needs N %% 4 == 0");
            return launch_gemm<ESMK_EPI_RESIDUAL>(A, lda, W, M, N, K, bias, alpha, out, s);
        case ESMK_EPI_BIAS_GELU:
            if (!bias) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: bias+gelu epilogue needs a bias");
            return launch_gemm<ESMK_EPI_BIAS_GELU>(A, lda, W, M, N, K, bias, alpha, out, s);
        default: ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: unknown epilogue %d", epilogue);
    }
}
