// Persistent 256x256x64 bf16 MFMA GEMM for the trunk's hot shapes (bf16 / SwiGLU epilogues).
//
// Same per-tile machine as gemm256.hip (8 waves, four 16-MFMA blocks per K-tile, LDS fragment reads
// software-pipelined inside the wave, LDS-DMA 1.5 K-tiles ahead, two barriers per K-tile), but one workgroup per
// CU walks MANY output tiles and treats their K-tiles as ONE continuous stream: the DMA of the next tile's
// first two K-tiles is already in flight while the current tile finishes, so the fill latency of a tile (the
// ~1/3 of a 1152-deep tile's lifetime that a non-persistent launch loses to prologue + epilogue + dispatch)
// is hidden.  The epilogue therefore cannot stage through LDS (it holds the next tile's operands): outputs go
// out as 8-byte buffer stores straight from the accumulators.
//
// vmcnt accounting: per wave the vector-memory queue holds, in issue order, DMA groups
//     ... U0-2(g) [6 pieces]  U3(g) [2]  U0-2(g+1) [6]  U3(g+1) [2] ...
// and, after the last K-tile of an output tile, exactly NS epilogue stores (buffer stores with hardware
// bounds checking - never branched around - so the count is exact).  A wait that must retire group X uses
// vmcnt(number of ops issued after X): 8 in steady state, 8+NS for the three waits that have the previous
// tile's stores between X and the youngest group (P and Q of a tile's first K-tile, P of its second).
#include <stdlib.h>

#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

constexpr int TB = 256;
constexpr int BK = 64;
constexpr int UNIT = 128 * 128;
constexpr int KBUF = 4 * UNIT;
constexpr int STAGE_BYTES = 32 * 1024;            // SwiGLU epilogue staging: 4 KiB per wave, beside the K-tile ring
constexpr int LDS_BYTES = 2 * KBUF + STAGE_BYTES;  // 160 KiB = all of a CU's LDS

#define SEG_BARRIER()                        \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

#ifdef ESMK_STAMPS
__device__ unsigned long long g_pstamps[8 * 64];
#define PSTAMP(i)                                                                                  \
    do {                                                                                           \
        if (blockIdx.x == 100 && lane == 0 && (i) < 64) {                                          \
            unsigned long long t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            g_pstamps[wave * 64 + (i)] = t_;                                                       \
        }                                                                                          \
    } while (0)
#else
#define PSTAMP(i)
#endif

#define WAITV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const __bf16* __restrict__ A, int lda,
                                                          const __bf16* __restrict__ W, int ldw, int M, int N,
                                                          int nt, float alpha, void* __restrict__ out,
                                                          int n_ctile, int nblk, int gt) {
    static_assert(EPI == ESMK_EPI_BF16 || EPI == ESMK_EPI_SWIGLU, "persistent kernel: bf16 / SwiGLU epilogues only");
    constexpr int NS = (EPI == ESMK_EPI_SWIGLU) ? 8 : 32;  // epilogue stores per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x;
    const int n_ttile = nblk / n_ctile;
    const int OW = (EPI == ESMK_EPI_SWIGLU) ? (N >> 1) : N;

    auto tile_coords = [&](int b, int& c0, int& t0) {
        const int id = xcd_remap(b, nblk);
        const int panel = id / (gt * n_ctile);
        const int gtp = min(gt, n_ttile - panel * gt);
        const int idp = id - panel * gt * n_ctile;
        const int ctile = idp / gtp;
        c0 = ctile * TB;
        t0 = (panel * gt + (idp - ctile * gtp)) * TB;
    };

    // DMA sources.  Wave w moves pieces 2w, 2w+1 (8 rows x 128 B) of every unit, in the buffer form of the LDS-DMA
    // (as gemm256.hip): tile base in a scalar descriptor, per-lane 32-bit offset, and the K offset plus the unit's
    // wave-uniform row step (unit 3: +64 W rows, unit 2: +32 A rows) in the scalar offset - four offset registers,
    // no per-piece 64-bit vector arithmetic, no branches.  Only wave-uniform tile state (the two bases) is carried
    // across K-tiles.  Token tiles are always full: esmk_gemm256p takes M % 256 == 0 only.
    struct Src {
        const char* wbase;
        const char* abase;
    };
    auto make_src = [&](int c0, int t0) {
        Src s;
        s.wbase = (const char*)(W + (size_t)c0 * ldw);
        s.abase = (const char*)(A + (size_t)t0 * lda);
        return s;
    };
    const int dma_dst = wave * 2048;
    unsigned offW[2], offA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 16 + i * 8 + (lane >> 3);
        const unsigned lc = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
        offW[i] = (unsigned)((r >> 6) * 128 + (r & 63)) * (unsigned)(ldw * 2) + lc;
        offA[i] = (unsigned)((r >> 5) * 64 + (r & 31)) * (unsigned)(lda * 2) + lc;
    }
    // stream position g -> LDS buffer g&1;  (src, kt) name the tile and K offset of that position
    auto issue_piece = [&](int u, int i, int g, const Src& s, int kt) {
        char* dst = smem + (g & 1) * KBUF + u * UNIT + dma_dst + i * 1024;
        const bool w = (u == 0 || u == 3);
        const int sof = kt * (BK * 2) + (u == 3 ? 64 * ldw * 2 : (u == 2 ? 32 * lda * 2 : 0));
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc((void*)(w ? s.wbase : s.abase), 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(dst), 16, (int)(w ? offW[i] : offA[i]), sof, 0, 0);
    };

    const int ra = wm * 64 + (lane & 15), rb = wn * 32 + (lane & 15);
    const int a_off = ra * 128 + (((lane >> 4) ^ ((ra >> 1) & 7)) << 4);
    const int b_off = rb * 128 + (((lane >> 4) ^ ((rb >> 1) & 7)) << 4);

    f32x4 acc[2][2][4][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    bf16x8 af[4][2], bfr[2][2][2];
    auto read_a_kk = [&](const char* unit, int kk) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi][kk] = *(const bf16x8*)(unit + ((a_off + mi * 2048) ^ (kk << 6)));
    };
    auto read_b_kk = [&](const char* buf, int kk) {
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bfr[qn][ni][kk] = *(const bf16x8*)(buf + (1 + qn) * UNIT + ((b_off + ni * 2048) ^ (kk << 6)));
    };

    // output buffer descriptor: stores past the end (or of masked lanes, offset forced out of range) are dropped
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)M * OW * 2), 0x00020000);
    auto epilogue = [&](int c0, int t0) {
        const int q4 = (lane >> 4) * 4;
        if constexpr (EPI == ESMK_EPI_SWIGLU) {
            // The wave's 64 tokens x 64 output channels go out as whole 128-byte rows: two passes of 32 tokens through
            // the wave's own 4 KiB of LDS BESIDE the K-tile ring (which already holds the next tile's operands).
            // Image: 32 rows x 128 B, 16-byte chunk XOR-swizzled by (row>>1)&7 (conflict-free 8-byte writes of 16 rows
            // x 32 B and 16-byte row reads).  DS operations of one wave execute in order: no waits between the passes.
            char* stg = smem + 2 * KBUF + wave * 4096;
            const int rrow = lane >> 3, rchunk = lane & 7;  // read side: 8 rows x 8 chunks per instruction
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int r = ni * 16 + (lane & 15);
#pragma unroll
                    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                        for (int mi = 0; mi < 4; mi += 2) {
                            const f32x4 g = acc[qm][qn][mi][ni], u = acc[qm][qn][mi + 1][ni];
                            u32x2 v;
                            v.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
                            v.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
                            const int cl = qm * 32 + (mi >> 1) * 16 + q4;  // channel inside the wave's 64
                            *(u32x2*)(stg + r * 128 + ((((cl >> 3) ^ (r >> 1)) & 7) << 4) + (cl & 4) * 2) = v;
                        }
                }
                const int cc = ((c0 + wm * 128) >> 1) + rchunk * 8;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + rrow;
                    const u32x4 v = *(const u32x4*)(stg + r * 128 + (((rchunk ^ (r >> 1)) & 7) << 4));
                    const int t = t0 + wn * 64 + qn * 32 + r;
                    const unsigned off = (t < M && cc < OW) ? (unsigned)t * (unsigned)(OW * 2) + (unsigned)cc * 2u : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (int)off, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int t = t0 + wn * 64 + qn * 32 + ni * 16 + (lane & 15);
                const unsigned rowoff = (t < M) ? (unsigned)t * (unsigned)(OW * 2) : 0x80000000u;
#pragma unroll
                for (int qm = 0; qm < 2; ++qm) {
                    const int cb = c0 + wm * 128 + qm * 64;
                    if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                        for (int mi = 0; mi < 4; mi += 2) {
                            const int c = (cb >> 1) + (mi >> 1) * 16 + q4;
                            const f32x4 g = acc[qm][qn][mi][ni], u = acc[qm][qn][mi + 1][ni];
                            u32x2 v;
                            v.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
                            v.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
                            const unsigned off = (c < OW) ? rowoff + (unsigned)c * 2u : 0x80000000u;
                            __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, (int)off, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            const int c = cb + mi * 16 + q4;
                            const f32x4 a4 = acc[qm][qn][mi][ni] * alpha;
                            u32x2 v;
                            v.x = pack_bf16x2(a4[0], a4[1]);
                            v.y = pack_bf16x2(a4[2], a4[3]);
                            const unsigned off = (c < OW) ? rowoff + (unsigned)c * 2u : 0x80000000u;
                            __builtin_amdgcn_raw_buffer_store_b64(v, orsrc, (int)off, 0, 0);
                        }
                    }
                }
            }
    };

#define MF(QM, KK, QN, MI, NI)                                                                     \
    acc[QM][QN][MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[MI][KK], bfr[QN][NI][KK], acc[QM][QN][MI][NI], 0, 0, 0);
#define MFPAIR(QM, KK, QN, MI, S)                                                                  \
    MF(QM, KK, QN, MI, 0) MF(QM, KK, QN, MI, 1)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    S;                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA_BLOCK(QM, KK, S0, S1, S2, S3, S4, S5, S6, S7)                                         \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    MFPAIR(QM, KK, 0, 0, S0) MFPAIR(QM, KK, 0, 1, S1) MFPAIR(QM, KK, 0, 2, S2) MFPAIR(QM, KK, 0, 3, S3) \
    MFPAIR(QM, KK, 1, 0, S4) MFPAIR(QM, KK, 1, 1, S5) MFPAIR(QM, KK, 1, 2, S6) MFPAIR(QM, KK, 1, 3, S7) \
    __builtin_amdgcn_s_setprio(0);
#define NOP_ ((void)0)

    // One K-tile at stream position g (M1: g+1 exists, M2: g+2 exists); (s2, k2) = tile / K index of g+2.
    // WP / WQ: statements that retire U3(g) before barrier P, resp. U0-2(g+1) before barrier Q.
#define KTILE(M1, M2, WP, WQ)                                                                      \
    {                                                                                              \
        const char* cur = smem + (g & 1) * KBUF;                                                   \
        const char* nxt = smem + ((g + 1) & 1) * KBUF;                                             \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + 0 * UNIT, 1); read_b_kk(cur, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        WP;                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + 3 * UNIT, 0); },                                        \
                   { if (M2) issue_piece(0, 0, g + 2, s2, k2); }, { if (M2) issue_piece(0, 1, g + 2, s2, k2); }, \
                   { if (M2) issue_piece(1, 0, g + 2, s2, k2); }, { if (M2) issue_piece(1, 1, g + 2, s2, k2); }, \
                   { if (M2) issue_piece(2, 0, g + 2, s2, k2); }, { if (M2) issue_piece(2, 1, g + 2, s2, k2); }, NOP_) \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + 3 * UNIT, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        WQ;                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(1, 1, { if (M1) { read_a_kk(nxt + 0 * UNIT, 0); read_b_kk(nxt, 0); } }, NOP_,   \
                   { if (M2) issue_piece(3, 0, g + 2, s2, k2); }, NOP_, { if (M2) issue_piece(3, 1, g + 2, s2, k2); }, NOP_, NOP_, NOP_) \
    }

    // ---- this workgroup's tiles: b = blockIdx.x, blockIdx.x + G, ... ----
    int b = blockIdx.x;
    const int my_tiles = (nblk - b + G - 1) / G;  // >= 1 (grid <= nblk)
    const int S = my_tiles * nt;                  // stream length in K-tiles; nt >= 2
    int c0, t0;
    tile_coords(b, c0, t0);
    Src scur = make_src(c0, t0);
    Src snext = scur;
    int nc0 = c0, nt0 = t0;
    bool has_next = (b + G < nblk);
    if (has_next) {
        tile_coords(b + G, nc0, nt0);
        snext = make_src(nc0, nt0);
    }

    // prologue: stream positions 0 and 1 (both in the first tile since nt >= 2)
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            issue_piece(u, 0, gg, scur, gg);
            issue_piece(u, 1, gg, scur, gg);
        }
        issue_piece(3, 0, gg, scur, gg);
        issue_piece(3, 1, gg, scur, gg);
    }
    WAITV(10);
    SEG_BARRIER();
    read_a_kk(smem + 0 * UNIT, 0);
    read_b_kk(smem, 0);

    int kt = 0;            // K index inside the current tile
    bool stored = false;   // the previous tile's NS stores sit in the queue (affects three waits)
    int g = 0;
    for (; g < S - 2; ++g) {
        // source of stream position g+2
        const bool in_cur = (kt + 2 < nt);
        const Src& s2 = in_cur ? scur : snext;
        const int k2 = in_cur ? kt + 2 : kt + 2 - nt;
        const int late = stored ? kt : 2;  // 0 / 1: the previous tile's stores are still ahead of the group we need
        KTILE(true, true,
              { if (late < 2) { if constexpr (NS == 8) WAITV(16); else WAITV(40); } else WAITV(8); },
              { if (late < 1) { if constexpr (NS == 8) WAITV(16); else WAITV(40); } else WAITV(8); })
        if (kt == 0) PSTAMP(3 * (g / nt));
        if (++kt == nt) {  // tile finished: flush it, move on (its successor's first two K-tiles are in flight)
            PSTAMP(3 * (g / nt) + 1);
            epilogue(c0, t0);
            PSTAMP(3 * (g / nt) + 2);
            zero_acc();
            stored = true;
            kt = 0;
            b += G;
            c0 = nc0; t0 = nt0;
            scur = snext;
            has_next = (b + G < nblk);
            if (has_next) {
                tile_coords(b + G, nc0, nt0);
                snext = make_src(nc0, nt0);
            }
        }
    }
    // last two K-tiles of the stream (always the last tile's, nt >= 2): conservative small counts
    {
        const Src& s2 = scur;
        const int k2 = 0;
        KTILE(true, false, WAITV(8), WAITV(2))   // P: U3(g) (newer: U0-2(g+1), U3(g+1) [+stores: over-wait is safe])
        ++g;
        KTILE(false, false, WAITV(0), NOP_)
    }
#undef KTILE
#undef MFMA_BLOCK
#undef MFPAIR
#undef MF
#undef NOP_
    epilogue(c0, t0);
}

template <int EPI>
int launch(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s) {
    const int Kpad = round_up(K, BK);
    const int n_ctile = round_up(N, TB) / TB;
    const int n_ttile = (M + TB - 1) / TB;
    const int nblk = n_ctile * n_ttile;
    static int gt = 0, ncu = 0;
    if (gt == 0) {
        gt = esmk_switch("ESMK_GEMM_GT", 4);
        if (gt < 1) gt = 1;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot query the device");
        ncu = prop.multiProcessorCount & ~7;  // multiple of 8 keeps a workgroup's tiles on one XCD
        if (ncu < 8) ncu = 8;
    }
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm256p_kernel<EPI>, LDS_BYTES, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", LDS_BYTES);
    const int grid = nblk < ncu ? nblk : ncu;
    hipLaunchKernelGGL(gemm256p_kernel<EPI>, dim3(grid), dim3(512), LDS_BYTES, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, alpha, out, n_ctile, nblk, gt);
    ESMK_CHECK_LAUNCH("esmk_gemm(256p)");
    return ESMK_OK;
}

}  // namespace

// Persistent path: bf16 / SwiGLU epilogues, K >= 128, M a multiple of 256, output < 2 GiB.  Returns 1 when the shape
// is not eligible.
int esmk_gemm256p(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, float alpha, void* out,
                  hipStream_t s) {
    const int OW = (epilogue == ESMK_EPI_SWIGLU) ? N / 2 : N;
    if (round_up(K, BK) / BK < 2 || (OW & 3) || (size_t)M * OW * 2 >= 0x7fffffffull) return 1;
    if (epilogue == ESMK_EPI_SWIGLU && (OW & 7)) return 1;  // staged epilogue stores 16-byte pieces of whole rows
    if (M % TB) return 1;                                   // full token tiles only (no row clamp in the DMA offsets)
    if (epilogue == ESMK_EPI_BF16) return launch<ESMK_EPI_BF16>(A, lda, W, M, N, K, alpha, out, s);
    if (epilogue == ESMK_EPI_SWIGLU) return launch<ESMK_EPI_SWIGLU>(A, lda, W, M, N, K, alpha, out, s);
    return 1;
}

#ifdef ESMK_STAMPS
extern "C" int esmk_debug_pstamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pstamps), n * sizeof(unsigned long long));
}
#endif
