// 256x256x64 bf16 MFMA GEMM: 8 waves, LDS-DMA 1.5 K-tiles ahead, LDS fragment reads software-pipelined
// inside every wave so the MFMA stream never waits for LDS; two workgroup barriers per K-tile.
//
// Same orientation and epilogues as gemm.hip (MFMA M = output channel, MFMA N = token).
// Wave (wm, wn) = (wave>>2, wave&3) owns channels wm*128..+127 x tokens wn*64..+63 = four 64x32 quadrants
// (qm, qn); a K-tile (BK = 64) is 64 v_mfma_f32_16x16x32_bf16 per wave in four blocks (qm, kk).
//
// LDS: 2 K-tile buffers x 4 UNITS of 16 KiB (128 rows x 128 B), a unit = the rows one block family reads:
//   U0 = A rows of qm=0 (both wm), U1 = B rows of qn=0 (all wn), U2 = B rows of qn=1, U3 = A rows of qm=1.
// Rows are 128 B with the 16-byte chunk XOR-swizzled by (row>>1)&7 (DMA source side + ds_read side): the
// 16-lane ds_read_b128 groups are conflict-free.  The schedule (who reads / overwrites which unit when, and
// why vmcnt(8) is the right count) is documented at the KTILE macro below.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

constexpr int TB = 256;             // tile edge (channels and tokens)
constexpr int BK = 64;
constexpr int UNIT = 128 * 128;     // bytes
constexpr int KBUF = 4 * UNIT;      // one K-tile buffer
constexpr int LDS_BYTES = 8 * 64 * (256 + 16);  // 136 KiB: max(K-loop buffers 128 KiB, epilogue staging)

#ifdef ESMK_STAMPS
__device__ unsigned long long g_stamps[8 * 64];
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (stamp_on && kt == 5 && lane == 0) {                                                    \
            unsigned long long t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            g_stamps[wave * 64 + (i)] = t_;                                                        \
        }                                                                                          \
    } while (0)
#else
#define STAMP(i)
#endif

#ifdef ESMK_NOBAR  /* timing experiment only: results are wrong */
#define SEG_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define SEG_BARRIER()                        \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)
#endif

constexpr int EPI_SWIGLU_BIAS = 102;  // internal: ESMK_EPI_SWIGLU with ffn.1.bias; its own instantiation, so the
                                      // bias-free kernel of the ESM C / ESM3 trunks is unchanged
constexpr int EPI_BF16_STATS = 103;   // internal: ESMK_EPI_BF16 of the QKV projection that also emits the q-side QK-LayerNorm
                                      // partial sums (gemm288.hip has the same for its tile widths): every wave sums its fp32
                                      // outputs per token over each 64-channel block (128-channel block when d_model > 2048,
                                      // so that a token has at most 32 parts) while they pass through the epilogue, and blocks
                                      // below stat_cols (= d_model: the q columns) store stats[token][block] = (sum, sum of squares)
template <int EPI_>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const __bf16* __restrict__ A, int lda,
                                                         const __bf16* __restrict__ W, int ldw, int M, int N,
                                                         int ktiles, const float* __restrict__ bias, float alpha,
                                                         void* __restrict__ out, int n_ctile, int nblk, int gt,
                                                         float* __restrict__ stats, int stat_cols) {
    constexpr bool kSwigluBias = (EPI_ == EPI_SWIGLU_BIAS);
    constexpr bool kStats = (EPI_ == EPI_BF16_STATS);
    constexpr int EPI = kSwigluBias ? ESMK_EPI_SWIGLU : kStats ? ESMK_EPI_BF16 : EPI_;
#ifdef ESMK_GEMM_DBG
    constexpr int dbg = ESMK_GEMM_DBG;  // diagnostic builds only: 1 = no DMA, 2 = no LDS reads, 4 = no epilogue
#else
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Tile order: each XCD walks a contiguous id range (xcd_remap); ids run through PANELS of `gt` token
    // tiles x all channel tiles with the token tile fastest, so the ~32 workgroups an XCD runs at once form a
    // (32/gt channel tiles) x (gt token tiles) super-tile: every A slice is shared by 32/gt blocks and every W
    // slice by gt blocks out of the XCD's L2 instead of the Infinity Cache / HBM.
    const int id = xcd_remap(blockIdx.x, nblk);
    const int n_ttile = nblk / n_ctile;
    const int panel = id / (gt * n_ctile);
    const int gtp = min(gt, n_ttile - panel * gt);  // last panel may be narrower
    const int idp = id - panel * gt * n_ctile;
    const int ctile = idp / gtp, ttile = panel * gt + (idp - ctile * gtp);
    const int c0 = ctile * TB, t0 = ttile * TB;
#ifdef ESMK_STAMPS
    const bool stamp_on = (blockIdx.x == 777);
#endif

    // ---- DMA sources: wave w moves pieces 2w, 2w+1 (8 rows x 128 B) of every unit.  Per-lane 32-bit byte
    //      offsets from two wave-uniform bases (W tile rows, A tile rows) keep the address math scalar. ----
    const char* wbase = (const char*)(W + (size_t)c0 * ldw);
    const char* abase = (const char*)(A + (size_t)t0 * lda);
    unsigned soff[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + (lane >> 3);          // unit row 0..127
        const int lc = ((lane & 7) ^ ((r >> 1) & 7)) * 16;       // logical chunk byte offset in the 128-B row
        const int arow = (r >> 6) * 128 + (r & 63);              // A row for qm = 0 (+64 for qm = 1)
        const int brow = (r >> 5) * 64 + (r & 31);               // B row for qn = 0 (+32 for qn = 1)
        int tr0 = brow, tr1 = brow + 32;
        tr0 = t0 + tr0 < M ? tr0 : M - 1 - t0;
        tr1 = t0 + tr1 < M ? tr1 : M - 1 - t0;
        soff[0][i] = (unsigned)arow * (unsigned)(ldw * 2) + lc;
        soff[1][i] = (unsigned)tr0 * (unsigned)(lda * 2) + lc;
        soff[2][i] = (unsigned)tr1 * (unsigned)(lda * 2) + lc;
        soff[3][i] = (unsigned)(arow + 64) * (unsigned)(ldw * 2) + lc;
    }
    const int dma_dst = wave * 2048;  // byte offset of this wave's two pieces inside a unit

    // ---- fragment read offsets inside a unit (kk = 0, mi/ni = 0) ----
    const int ra = wm * 64 + (lane & 15), rb = wn * 32 + (lane & 15);
    const int a_off = ra * 128 + (((lane >> 4) ^ ((ra >> 1) & 7)) << 4);
    const int b_off = rb * 128 + (((lane >> 4) ^ ((rb >> 1) & 7)) << 4);

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[2][2][2];  // A sub-tile of the current qm; B sub-tiles of BOTH qn

    // buffer form of the LDS-DMA (buffer_load_dwordx4 ... offen lds): tile base in a scalar descriptor, per-lane
    // 32-bit offset, K offset in an SGPR - no per-piece 64-bit vector address arithmetic in the loop (2-3 % faster
    // than global_load_lds with a per-lane 64-bit address, measured on the 288-tile kernel)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    auto issue_piece = [&](int u, int i, int kt) {
        if (dbg & 1) return;
        char* dst = smem + (kt & 1) * KBUF + u * UNIT + dma_dst + i * 1024;
#ifdef ESMK_SAMEDMA  /* timing experiment only: every tile streams the same 64 KiB */
        const int koff = 0;
#else
        const int koff = kt * (BK * 2);
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds((u == 0 || u == 3) ? rW : rA, LDS_PTR(dst), 16, (int)soff[u][i], koff, 0, 0);
    };
    auto issue_unit = [&](int u, int kt) {
        issue_piece(u, 0, kt);
        issue_piece(u, 1, kt);
    };
    auto read_a_kk = [&](const char* unit, int kk) {
        if (dbg & 2) return;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi][kk] = *(const bf16x8*)(unit + ((a_off + mi * 2048) ^ (kk << 6)));
    };
    auto read_b_kk = [&](const char* buf, int kk) {
        if (dbg & 2) return;
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bfr[qn][ni][kk] = *(const bf16x8*)(buf + (1 + qn) * UNIT + ((b_off + ni * 2048) ^ (kk << 6)));
    };

    // ---- prologue: U0,U1,U2,U3 of tile 0 and of tile 1; wait for U0..U2(0); load the kk=0 fragments ----
    issue_unit(0, 0);
    issue_unit(1, 0);
    issue_unit(2, 0);
    issue_unit(3, 0);
    if (ktiles > 1) {
        issue_unit(0, 1);
        issue_unit(1, 1);
        issue_unit(2, 1);
        issue_unit(3, 1);
#ifdef ESMK_G256_ONEBAR
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // all of tile 0 (block 2 reads U3 without a barrier of its own)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#else
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
#endif
    SEG_BARRIER();
    read_a_kk(smem + 0 * UNIT, 0);
    read_b_kk(smem, 0);

    // 16 MFMAs of block (QM, KK) as 8 pairs; S0..S7 are statements issued behind pair 0..7.  LDS fragment reads
    // for a later block go behind pair 0 (the compiler's lgkmcnt(0) in front of the first MFMA then only covers
    // reads issued a whole block earlier); LDS-DMA pieces are spread one per pair so that a full vector-memory
    // queue never blocks the wave's MFMA stream for long.
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

    // One K-tile = four blocks of 16 MFMAs; every block also issues the LDS reads of fragments that a LATER
    // block consumes, into registers the PREVIOUS block retired (software pipeline inside the wave):
    //   block 1 (qm0,kk0): reads kk1 of A(qm0)+B            then  vmcnt / lgkmcnt(0) / barrier P
    //   block 2 (qm0,kk1): reads kk0 of A(qm1); DMA U0-2(kt+2), one piece per MFMA pair
    //   block 3 (qm1,kk0): reads kk1 of A(qm1)              then  vmcnt / lgkmcnt(0) / barrier Q
    //   block 4 (qm1,kk1): reads kk0 of A(qm0)+B of tile kt+1; DMA U3(kt+2)
    // Barrier P(kt): U3(kt) has landed for every wave, and every wave is done with slots U0-2 of this buffer.
    // Barrier Q(kt): U0-2(kt+1) have landed, and every wave is done with slot U3 of this buffer.
    // Issue order per wave is U0-2(t), U3(t), U0-2(t+1), U3(t+1), ... (6 + 2 pieces), so "all but the two
    // youngest groups" = s_waitcnt vmcnt(8).  M1: tile kt+1 exists, M2: tile kt+2 exists.
#ifdef ESMK_G256_ONEBAR
    // Experiment (round 3), NOT the default: ONE workgroup barrier per K-tile instead of two.  Bit-identical; back to
    // back on resident operands it is 2-5 % faster (FFN up-projection 354 -> 339..346 us, 8192 x 8192 x 1536 +5 %), but INSIDE
    // the forward, where the activations come from beyond L2, it is slower (FFN up-projection 371.8 -> 377 us in situ,
    // 40.30 -> 40.45 ms per step): with two buffers one barrier means ONE K-tile of flight instead of 1.5.  All eight pieces of tile kt+2 are issued
    // behind barrier S(kt) (into the buffer tile kt has just finished with) during block 4, one per MFMA pair: one
    // K-tile of flight instead of 1.5, half the rendezvous.  S(kt): every piece of tile kt+1 has landed for every wave
    // (nothing younger is outstanding: vmcnt(0)), and every wave has issued its last read of buffer kt (block 3).
#define KTILE(M1, M2)                                                                              \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * KBUF;                                                  \
        const char* nxt = smem + ((kt + 1) & 1) * KBUF;                                            \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + 0 * UNIT, 1); read_b_kk(cur, 1); }, NOP_,               \
                   { if (M1 && kt > 0) issue_piece(3, 1, kt + 1); }, NOP_, NOP_, NOP_, NOP_, NOP_)   \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + 3 * UNIT, 0); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + 3 * UNIT, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(1, 1, { if (M1) { read_a_kk(nxt + 0 * UNIT, 0); read_b_kk(nxt, 0); } },         \
                   { if (M2) issue_piece(0, 0, kt + 2); }, { if (M2) issue_piece(0, 1, kt + 2); }, \
                   { if (M2) issue_piece(1, 0, kt + 2); }, { if (M2) issue_piece(1, 1, kt + 2); }, \
                   { if (M2) issue_piece(2, 0, kt + 2); }, { if (M2) issue_piece(2, 1, kt + 2); }, \
                   { if (M2) issue_piece(3, 0, kt + 2); })                                         \
    }
#else
#define KTILE(M1, M2)                                                                              \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * KBUF;                                                  \
        const char* nxt = smem + ((kt + 1) & 1) * KBUF;                                            \
        STAMP(0);                                                                                  \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + 0 * UNIT, 1); read_b_kk(cur, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        if (M1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        STAMP(1);                                                                                  \
        SEG_BARRIER();                                                                             \
        STAMP(2);                                                                                  \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + 3 * UNIT, 0); },                                        \
                   { if (M2) issue_piece(0, 0, kt + 2); }, { if (M2) issue_piece(0, 1, kt + 2); }, \
                   { if (M2) issue_piece(1, 0, kt + 2); }, { if (M2) issue_piece(1, 1, kt + 2); }, \
                   { if (M2) issue_piece(2, 0, kt + 2); }, { if (M2) issue_piece(2, 1, kt + 2); }, NOP_) \
        STAMP(3);                                                                                  \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + 3 * UNIT, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        if (M2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                   \
        else if (M1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        STAMP(4);                                                                                  \
        SEG_BARRIER();                                                                             \
        STAMP(5);                                                                                  \
        MFMA_BLOCK(1, 1, { if (M1) { read_a_kk(nxt + 0 * UNIT, 0); read_b_kk(nxt, 0); } }, NOP_,   \
                   { if (M2) issue_piece(3, 0, kt + 2); }, NOP_, { if (M2) issue_piece(3, 1, kt + 2); }, NOP_, NOP_, NOP_) \
        STAMP(6);                                                                                  \
    }

#endif
    int kt = 0;
    for (; kt < ktiles - 2; ++kt) KTILE(true, true)
    if (ktiles >= 2) {
        KTILE(true, false)
        ++kt;
    }
    KTILE(false, false)
#undef KTILE
#undef MFMA_BLOCK
#undef MFPAIR
#undef MF
#undef NOP_

    // ---- epilogue ----
    if (dbg & 4) {
        if (acc[0][0][0][0][0] == 123.456f) ((float*)out)[0] = acc[1][1][3][1][2];
        return;
    }
    const int q4 = (lane >> 4) * 4;
    constexpr bool kBf16Out = (EPI == ESMK_EPI_BF16 || EPI == ESMK_EPI_SWIGLU || EPI == ESMK_EPI_BIAS_GELU);
    const int OW = (EPI == ESMK_EPI_SWIGLU) ? (N >> 1) : N;  // output width
    if (kBf16Out && (OW & 7) == 0) {
        // bf16 outputs: stage this wave's 64-token x CW-channel block through its own LDS region (the K-loop is
        // done with LDS: no wave reads fragments after barrier Q of the last tile) and store whole rows,
        // 16 B per lane, CW*2 contiguous bytes per token.  No inter-wave synchronisation needed.
        constexpr int CW = (EPI == ESMK_EPI_SWIGLU) ? 64 : 128;     // channels per wave after the epilogue op
        constexpr int PITCH = CW * 2 + 16;                          // row pitch in bytes (+16: bank spread)
        char* stg = smem + wave * (64 * PITCH);
        float ssum[kStats ? 2 : 1][2][2], ssq[kStats ? 2 : 1][2][2];  // kStats: by (qm, qn, ni), this lane's four channel quads
        if constexpr (kStats) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) ssum[a][b][c] = ssq[a][b][c] = 0.f;
        }
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int trow = qn * 32 + ni * 16 + (lane & 15);
                    if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                        for (int mi = 0; mi < 4; mi += 2) {
                            f32x4 g = acc[qm][qn][mi][ni], u = acc[qm][qn][mi + 1][ni];
                            if constexpr (kSwigluBias) {
                                const int c = ((c0 + wm * 128) >> 1) + qm * 32 + (mi >> 1) * 16 + q4;  // output channel
                                if (c < OW) {
                                    g += *(const f32x4*)(bias + c);
                                    u += *(const f32x4*)(bias + OW + c);
                                }
                            }
                            u32x2 v;
                            v.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
                            v.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
                            *(u32x2*)(stg + trow * PITCH + (qm * 32 + (mi >> 1) * 16 + q4) * 2) = v;
                        }
                    } else {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            f32x4 v = acc[qm][qn][mi][ni];
                            const int cl = qm * 64 + mi * 16 + q4;
                            if constexpr (EPI == ESMK_EPI_BIAS_GELU) {
                                const int c = c0 + wm * 128 + cl;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r] + (c + r < N ? bias[c + r] : 0.f));
                            } else {
                                v = scale4(v, alpha);
                            }
                            if constexpr (kStats) {  // summed while the values pass through (a sweep of its own over the
                                                     // accumulators doubles their live range)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    ssum[qm][qn][ni] += v[r];
                                    ssq[qm][qn][ni] += v[r] * v[r];
                                }
                            }
                            u32x2 pk;
                            pk.x = pack_bf16x2(v[0], v[1]);
                            pk.y = pack_bf16x2(v[2], v[3]);
                            *(u32x2*)(stg + trow * PITCH + cl * 2) = pk;
                        }
                    }
                }
        if constexpr (kStats) {
            const bool g128 = stat_cols > 2048;                 // 128-channel parts: the two qm blocks of the wave together
            const int parts = stat_cols >> (g128 ? 7 : 6);
            const int cbw = c0 + wm * 128;                      // first channel of this wave (wave-uniform)
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                if (g128 && qm) break;
                const int cb = cbw + qm * 64;
                if (cb >= stat_cols) break;                     // k and v columns: no statistics
#pragma unroll
                for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        // the four 16-lane groups hold four different channel quads of the same 16 tokens
                        float sm = ssum[qm][qn][ni], sq = ssq[qm][qn][ni];
                        if (g128) {
                            sm += ssum[1][qn][ni];
                            sq += ssq[1][qn][ni];
                        }
                        sm += __shfl_xor(sm, 16);
                        sq += __shfl_xor(sq, 16);
                        sm += __shfl_xor(sm, 32);
                        sq += __shfl_xor(sq, 32);
                        const int t = t0 + wn * 64 + qn * 32 + ni * 16 + (lane & 15);
                        if (lane < 16 && t < M) {
                            float* sp = stats + ((size_t)t * parts + (g128 ? cb >> 7 : cb >> 6)) * 2;
                            sp[0] = sm;
                            sp[1] = sq;
                        }
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int LPR = CW / 8;         // lanes per row (16 B each)
        constexpr int RPI = 64 / LPR;       // rows per wave-instruction
        const int cbase = (EPI == ESMK_EPI_SWIGLU) ? ((c0 + wm * 128) >> 1) : (c0 + wm * 128);
        const int cc = cbase + (lane % LPR) * 8;
#pragma unroll
        for (int it = 0; it < 64 / RPI; ++it) {
            const int trow = it * RPI + lane / LPR;
            const int t = t0 + wn * 64 + trow;
            const u32x4 v = *(const u32x4*)(stg + trow * PITCH + (lane % LPR) * 16);
            if (t < M && cc < OW) *(u32x4*)((__bf16*)out + (size_t)t * OW + cc) = v;
        }
        return;
    }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int t = t0 + wn * 64 + qn * 32 + ni * 16 + (lane & 15);
            if (t >= M) continue;
#pragma unroll
            for (int qm = 0; qm < 2; ++qm) {
                const int cb = c0 + wm * 128 + qm * 64;
                if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                    for (int mi = 0; mi < 4; mi += 2)
                        epi_swiglu4(acc[qm][qn][mi][ni], acc[qm][qn][mi + 1][ni], t, (cb >> 1) + (mi >> 1) * 16 + q4,
                                    N >> 1, out, kSwigluBias ? bias : nullptr);
                } else {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        epi_store4<EPI>(acc[qm][qn][mi][ni], t, cb + mi * 16 + q4, N, bias, alpha, out);
                }
            }
        }
}

template <int EPI>
int launch(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, float alpha, void* out,
           hipStream_t s, int n_ctile, float* stats = nullptr, int stat_cols = 0) {
    const int Kpad = round_up(K, BK);
    const int n_ttile = (M + TB - 1) / TB;
    const int nblk = n_ctile * n_ttile;
    int gt = esmk_switch("ESMK_GEMM_GT", 4);  // panel width of the tile order (4: best or within noise on every shape)
    if (gt < 1) gt = 1;
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm256_kernel<EPI>, LDS_BYTES, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", (int)(LDS_BYTES));
    hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(nblk), dim3(512), LDS_BYTES, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, bias, alpha, out, n_ctile, nblk, gt, stats, stat_cols);
    ESMK_CHECK_LAUNCH("esmk_gemm(256)");
    return ESMK_OK;
}

}  // namespace

// Called by esmk_gemm (gemm.hip) for large problems.  W must be packed with rows padded to 256.
// Computes the first n_ctile 256-channel tile columns (all of them, or all but the last under split dispatch).
int esmk_gemm256(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                 float alpha, void* out, hipStream_t s, int n_ctile) {
    switch (epilogue) {
        case ESMK_EPI_BF16: return launch<ESMK_EPI_BF16>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
        case ESMK_EPI_F32: return launch<ESMK_EPI_F32>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
        case ESMK_EPI_SWIGLU:
            if (bias) return launch<EPI_SWIGLU_BIAS>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
            return launch<ESMK_EPI_SWIGLU>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
        case ESMK_EPI_RESIDUAL: return launch<ESMK_EPI_RESIDUAL>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
        case ESMK_EPI_BIAS_GELU: return launch<ESMK_EPI_BIAS_GELU>(A, lda, W, M, N, K, bias, alpha, out, s, n_ctile);
        default: ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_gemm: unknown epilogue %d", epilogue);
    }
}

// QKV projection (N = 3 d, bf16 out, no bias) that also emits the q-side LayerNorm partial sums (EPI_BF16_STATS above):
// stats is [M][parts][2] with parts = d / 64 (d <= 2048) or d / 128.  The caller (gemm.hip) checks the shape.
int esmk_gemm256_stats(const void* A, int lda, const void* W, int M, int d, void* out, float* stats, hipStream_t s,
                       int n_ctile) {
    return launch<EPI_BF16_STATS>(A, lda, W, M, 3 * d, d, nullptr, 1.f, out, s, n_ctile, stats, d);
}

#ifdef ESMK_STAMPS
extern "C" int esmk_debug_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#endif
