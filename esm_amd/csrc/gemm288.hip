// 288(channel) x 256(token) x 64 bf16 MFMA GEMM for output widths that are multiples of 288 but not of 256
// (d_model = 1152: N = 1152 = 4 x 288 and 3 x 1152 = 12 x 288).  With 256-wide channel tiles those shapes waste a
// half-empty tile column and a partial round of workgroups (out-proj / FFN2: 640 tiles = 2.5 rounds of 256 CUs;
// here 512 tiles = 2 rounds; QKV: 14 columns -> 12).  bf16 output, out = alpha * A.W^T (ESMK_EPI_BF16) only.
//
// Same structure as gemm256.hip (read that file first): 8 waves = 2 (channel halves, 144 rows) x 4 (token
// quarters, 64 rows); LDS-DMA 1.5 K-tiles ahead with counted vmcnt; LDS fragment reads software-pipelined inside
// each wave; two workgroup barriers per K-tile.  A wave's 144 channels are THREE row groups qm of 48 (3 MFMA rows),
// so a K-tile is six blocks (qm, kk) of 12 MFMAs.
//
// LDS: 2 K-tile buffers x 68 KiB = [UB0 16K | UB1 16K | UA0 12K | UA1 12K | UA2 12K], 128-byte rows, 16-byte chunk
// XOR-swizzled by (row>>1)&7.  UBq = token rows of quadrant column qn (all wn), UAq = channel rows of group qm (both
// wm).  The DMA moves the P region [UB0|UB1|UA0] (44 pieces of 1 KiB) and the Q region [UA1|UA2] (24 pieces):
// wave w owns P pieces w, w+8, ... (6 for waves 0-3, 5 for waves 4-7: template HI) and Q pieces w, w+8, w+16.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

// NQM = channel row groups of 48 per wave: 3 -> 288-channel tiles, 2 -> 192-channel tiles (d_model 1536 at 8192
// tokens: 1536 = 8 x 192 gives exactly 256 workgroups per round where 256-wide tiles give 192)
constexpr int TT = 256, BK = 64;
constexpr int UB = 128 * 128, UA = 96 * 128;
constexpr int OFF_UB0 = 0, OFF_UB1 = UB, OFF_UA0 = 2 * UB, OFF_UA1 = 2 * UB + UA, OFF_UA2 = 2 * UB + 2 * UA;
template <int NQM> struct Cfg {
    static constexpr int TC = 96 * NQM;                    // channels per tile
    static constexpr int WC = 48 * NQM;                    // channels per wave
    static constexpr int KBUF = 2 * UB + NQM * UA;         // 69632 B (288) / 57344 B (192)
    static constexpr int PITCH = WC * 2 + 16;              // epilogue staging row pitch
    static constexpr int LDS_BYTES = (8 * 64 * PITCH > 2 * KBUF) ? 8 * 64 * PITCH : 2 * KBUF;
    static constexpr int LPR = WC / 8;                     // lanes per staged row (16 B each): 18 / 12
    static constexpr int RPI = 64 / LPR;                   // rows per wave instruction: 3 / 5
};

#define SEG_BARRIER()                        \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// STATS (QKV projection with the q-side QK-LayerNorm moved into the attention kernel): tiles whose channels lie below
// stat_cols (= d_model: the q columns) also write, per token and per wave half, the sum and the sum of squares of
// their WC fp32 outputs: stats[token][2 * ctile + wm] = (sum, sum of squares).
template <int NQM, bool HI, bool STATS>  // HI: waves 0-3 (six P pieces per K-tile), else five
__device__ __forceinline__ void gemm288_body(const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ W, int ldw,
                                             int M, int N, int ktiles, float alpha, __bf16* __restrict__ out, int c0,
                                             int t0, char* smem, float* __restrict__ stats, int stat_cols) {
    using C = Cfg<NQM>;
    constexpr int KBUF = C::KBUF, PITCH = C::PITCH;
    // Q region = UA1 (.. UA2): 12 (NQM - 1) pieces over 8 waves -> 3 per wave (288) / 2 for waves 0-3 and 1 else (192)
    constexpr int NP = HI ? 6 : 5, NQ = (NQM == 3) ? 3 : (HI ? 2 : 1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- DMA sources.  Piece p of a region covers its rows 8p..8p+7; this lane moves row 8p + (lane>>3),
    //      logical chunk (lane&7) ^ swizzle(row).  P region: pieces 0..31 are token rows (UB0, UB1), 32..43 UA0. ----
    const char* wbase = (const char*)(W + (size_t)c0 * ldw);
    const char* abase = (const char*)(A + (size_t)t0 * lda);
    auto a_row_off = [&](int unit_row, int qm) {  // channel row of UA[qm] -> byte offset from wbase
        const int w2 = unit_row / 48, rem = unit_row - w2 * 48;
        return (unsigned)(w2 * C::WC + qm * 48 + rem) * (unsigned)(ldw * 2);
    };
    auto b_row_off = [&](int unit_row, int qn) {  // token row of UB[qn] -> byte offset from abase (clamped to M-1)
        int tr = (unit_row >> 5) * 64 + qn * 32 + (unit_row & 31);
        tr = t0 + tr < M ? tr : M - 1 - t0;
        return (unsigned)tr * (unsigned)(lda * 2);
    };
    unsigned soffP[6], soffQ[3];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int p = wave + 8 * j;                       // piece inside the P region
        const int r8 = lane >> 3;
        if (j < 4) {                                      // UB0 (j = 0,1) / UB1 (j = 2,3)
            const int ur = (p & 15) * 8 + r8;
            soffP[j] = b_row_off(ur, j >> 1) + (((lane & 7) ^ ((ur >> 1) & 7)) << 4);
        } else {                                          // UA0: pieces 32..43
            const int ur = (p - 32) * 8 + r8;             // < 96 for the pieces that exist (HI: j = 5 -> p = 40..43)
            soffP[j] = a_row_off(ur < 96 ? ur : 95, 0) + (((lane & 7) ^ ((ur >> 1) & 7)) << 4);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int p = wave + 8 * j;                             // 0..23: UA1 = 0..11, UA2 = 12..23
        if (p >= 12 * (NQM - 1)) p = 12 * (NQM - 1) - 1;  // pieces this wave never issues (j >= NQ)
        const int qm = p < 12 ? 1 : 2;
        const int ur = (p - (qm - 1) * 12) * 8 + (lane >> 3);
        soffQ[j] = a_row_off(ur, qm) + (((lane & 7) ^ ((ur >> 1) & 7)) << 4);
    }
    // buffer form of the LDS-DMA: tile base in a scalar descriptor, per-lane 32-bit offset, K offset in an SGPR -
    // no per-piece 64-bit vector address arithmetic in the loop
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    auto issue_P = [&](int j, int kt) {  // j compile-time after unrolling
        char* dst = smem + (kt & 1) * KBUF + (wave + 8 * j) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(j < 4 ? rA : rW, LDS_PTR(dst), 16, (int)soffP[j], kt * (BK * 2), 0, 0);
    };
    auto issue_Q = [&](int j, int kt) {
        char* dst = smem + (kt & 1) * KBUF + OFF_UA1 + (wave + 8 * j) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, (int)soffQ[j], kt * (BK * 2), 0, 0);
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_P(j, kt);
#pragma unroll
        for (int j = 0; j < NQ; ++j) issue_Q(j, kt);
    };

    // ---- fragment read offsets inside a unit (kk = 0, mi / ni = 0) ----
    const int ra = wm * 48 + (lane & 15), rb = wn * 32 + (lane & 15);
    const int a_off = ra * 128 + (((lane >> 4) ^ ((ra >> 1) & 7)) << 4);
    const int b_off = rb * 128 + (((lane >> 4) ^ ((rb >> 1) & 7)) << 4);

    f32x4 acc[NQM][2][3][2];  // [qm][qn][mi][ni]
#pragma unroll
    for (int a = 0; a < NQM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[3][2], bfr[2][2][2];  // A rows of the current qm; B rows of both qn

    auto read_a_kk = [&](const char* unit, int kk) {
#pragma unroll
        for (int mi = 0; mi < 3; ++mi) af[mi][kk] = *(const bf16x8*)(unit + ((a_off + mi * 2048) ^ (kk << 6)));
    };
    auto read_b_kk = [&](const char* buf, int kk) {
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bfr[qn][ni][kk] = *(const bf16x8*)(buf + qn * UB + ((b_off + ni * 2048) ^ (kk << 6)));
    };

    // ---- prologue: tiles 0 and 1 in flight; wait for P(0) ----
    issue_tile(0);
    if (ktiles > 1) {
        issue_tile(1);
#ifdef ESMK_G288_ONEBAR
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NQ) : "memory");  // ALL of tile 0 (no barrier P certifies its Q region)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NQ + NP) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
    }
#endif
    SEG_BARRIER();
    read_a_kk(smem + OFF_UA0, 0);
    read_b_kk(smem, 0);

#define MF(QM, KK, QN, MI, NI)                                                                     \
    acc[QM][QN][MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[MI][KK], bfr[QN][NI][KK], acc[QM][QN][MI][NI], 0, 0, 0);
#define MFPAIR(QM, KK, QN, MI, S)                                                                  \
    MF(QM, KK, QN, MI, 0) MF(QM, KK, QN, MI, 1)                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    S;                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA_BLOCK(QM, KK, S0, S1, S2, S3, S4, S5)                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    MFPAIR(QM, KK, 0, 0, S0) MFPAIR(QM, KK, 0, 1, S1) MFPAIR(QM, KK, 0, 2, S2)                     \
    MFPAIR(QM, KK, 1, 0, S3) MFPAIR(QM, KK, 1, 1, S4) MFPAIR(QM, KK, 1, 2, S5)                     \
    __builtin_amdgcn_s_setprio(0);
#define NOP_ ((void)0)

    // Six blocks per K-tile; every block issues the LDS reads of the fragments the NEXT block consumes:
    //   B1 (qm0,kk0): reads kk1 of UA0 + UB            then vmcnt / lgkmcnt(0) / barrier P
    //   B2 (qm0,kk1): reads kk0 of UA1; DMA P pieces 0-4 of tile kt+2 (the P region of this buffer is free)
    //   B3 (qm1,kk0): reads kk1 of UA1; DMA P piece 5 (waves 0-3)
    //   B4 (qm1,kk1): reads kk0 of UA2
    //   B5 (qm2,kk0): reads kk1 of UA2                 then vmcnt / lgkmcnt(0) / barrier Q
    //   B6 (qm2,kk1): reads kk0 of UA0 + UB of tile kt+1; DMA Q pieces of tile kt+2
    // Barrier P(kt): Q(kt) (UA1, UA2) has landed for every wave; all waves are done with the P region of this buffer.
    // Barrier Q(kt): P(kt+1) has landed; all waves are done with UA1 / UA2 of this buffer.
    // Issue order per wave: P(t), Q(t), P(t+1), Q(t+1), ...  =>  "all but the two youngest groups" = vmcnt(NP + NQ).
#ifdef ESMK_G288_ONEBAR
    // Experiment, NOT the default: ONE workgroup barrier per K-tile as in gemm256.hip.  Bit-identical but slower on this
    // kernel's six-block K-tiles (FFN down-projection 170 -> 187 us, QKV -1 %): the 1.5 K-tiles of flight matter more here.
    // every piece of tile kt+2 is issued behind barrier
    // S(kt), into the buffer tile kt is done with - P pieces 0-4 in the last block of tile kt, the rest in the first
    // block of tile kt+1.  S(kt): all of tile kt+1 has landed for every wave (nothing younger outstanding: vmcnt(0)),
    // and every wave has issued its last read of buffer kt.
#define KTILE(M1, M2)                                                                              \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * KBUF;                                                  \
        const char* nxt = smem + ((kt + 1) & 1) * KBUF;                                            \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + OFF_UA0, 1); read_b_kk(cur, 1); },                      \
                   { if (M1 && kt > 0 && HI) issue_P(5, kt + 1); }, { if (M1 && kt > 0 && NQ > 0) issue_Q(0, kt + 1); }, \
                   { if (M1 && kt > 0 && NQ > 1) issue_Q(1, kt + 1); }, { if (M1 && kt > 0 && NQ > 2) issue_Q(2, kt + 1); }, NOP_) \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + OFF_UA1, 0); }, NOP_, NOP_, NOP_, NOP_, NOP_)           \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + OFF_UA1, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_)           \
        if constexpr (NQM == 3) {                                                                  \
            MFMA_BLOCK(1, 1, { read_a_kk(cur + OFF_UA2, 0); }, NOP_, NOP_, NOP_, NOP_, NOP_)       \
            MFMA_BLOCK(2, 0, { read_a_kk(cur + OFF_UA2, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_)       \
        }                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(NQM - 1, 1, { if (M1) { read_a_kk(nxt + OFF_UA0, 0); read_b_kk(nxt, 0); } },    \
                   { if (M2) issue_P(0, kt + 2); }, { if (M2) issue_P(1, kt + 2); }, { if (M2) issue_P(2, kt + 2); }, \
                   { if (M2) issue_P(3, kt + 2); }, { if (M2) issue_P(4, kt + 2); })               \
    }
#else
#define KTILE(M1, M2)                                                                              \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * KBUF;                                                  \
        const char* nxt = smem + ((kt + 1) & 1) * KBUF;                                            \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + OFF_UA0, 1); read_b_kk(cur, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_) \
        if (M1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NQ) : "memory");                     \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + OFF_UA1, 0); }, { if (M2) issue_P(0, kt + 2); },        \
                   { if (M2) issue_P(1, kt + 2); }, { if (M2) issue_P(2, kt + 2); },               \
                   { if (M2) issue_P(3, kt + 2); }, { if (M2) issue_P(4, kt + 2); })               \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + OFF_UA1, 1); }, { if (M2 && HI) issue_P(5, kt + 2); }, NOP_, NOP_, NOP_, NOP_) \
        if constexpr (NQM == 3) {                                                                  \
            MFMA_BLOCK(1, 1, { read_a_kk(cur + OFF_UA2, 0); }, NOP_, NOP_, NOP_, NOP_, NOP_)       \
            MFMA_BLOCK(2, 0, { read_a_kk(cur + OFF_UA2, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_)       \
        }                                                                                          \
        if (M2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NQ) : "memory");                     \
        else if (M1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(NQM - 1, 1, { if (M1) { read_a_kk(nxt + OFF_UA0, 0); read_b_kk(nxt, 0); } },    \
                   { if (M2 && NQ > 0) issue_Q(0, kt + 2); }, { if (M2 && NQ > 1) issue_Q(1, kt + 2); }, \
                   { if (M2 && NQ > 2) issue_Q(2, kt + 2); }, NOP_, NOP_)                          \
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

    // ---- epilogue: stage this wave's 64 tokens x 144 channels through its own LDS region (no wave reads
    //      fragments after barrier Q of the last tile) and store whole 288-byte row pieces, 16 B per lane ----
    const int q4 = (lane >> 4) * 4;
    char* stg = smem + wave * (64 * PITCH);
    float ssum[2][2], ssq[2][2];  // STATS: per-token partial sums of this lane's channels, by (qn, ni)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ssum[i][j] = ssq[i][j] = 0.f;
#pragma unroll
    for (int qm = 0; qm < NQM; ++qm)
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int trow = qn * 32 + ni * 16 + (lane & 15);
#pragma unroll
                for (int mi = 0; mi < 3; ++mi) {
                    const f32x4 v = scale4(acc[qm][qn][mi][ni], alpha);
                    if constexpr (STATS) {  // summed while the values pass through anyway (a separate sweep over the
                                            // accumulators doubles their live range and spills)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ssum[qn][ni] += v[e];
                            ssq[qn][ni] += v[e] * v[e];
                        }
                    }
                    u32x2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *(u32x2*)(stg + trow * PITCH + (qm * 48 + mi * 16 + q4) * 2) = pk;
                }
            }
    if constexpr (STATS) {
        if (c0 < stat_cols) {  // workgroup-uniform
            const int parts = 2 * (stat_cols / C::TC);
            const int part = 2 * (c0 / C::TC) + wm;
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    // the four 16-lane groups hold four different channel quads of the same 16 tokens
                    float sm = ssum[qn][ni], sq = ssq[qn][ni];
                    sm += __shfl_xor(sm, 16);
                    sq += __shfl_xor(sq, 16);
                    sm += __shfl_xor(sm, 32);
                    sq += __shfl_xor(sq, 32);
                    const int t = t0 + wn * 64 + qn * 32 + ni * 16 + (lane & 15);
                    if (lane < 16 && t < M) {
                        float* sp = stats + ((size_t)t * parts + part) * 2;
                        sp[0] = sm;
                        sp[1] = sq;
                    }
                }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // LPR lanes cover one row of WC channels (288 / 192 bytes): RPI rows per wave instruction
    constexpr int LPR = C::LPR, RPI = C::RPI;
    const int lr = lane / LPR, lc = lane - lr * LPR;
    const int cc = c0 + wm * C::WC + lc * 8;
#pragma unroll
    for (int it = 0; it < (64 + RPI - 1) / RPI; ++it) {
        const int trow = it * RPI + lr;
        const int t = t0 + wn * 64 + trow;
        if (lane < LPR * RPI && trow < 64 && t < M) {
            const u32x4 v = *(const u32x4*)(stg + trow * PITCH + lc * 16);
            *(u32x4*)(out + (size_t)t * N + cc) = v;
        }
    }
}

template <int NQM, bool STATS>
__global__ __launch_bounds__(512, 2) void gemm288_kernel(const __bf16* __restrict__ A, int lda,
                                                         const __bf16* __restrict__ W, int ldw, int M, int N,
                                                         int ktiles, float alpha, __bf16* __restrict__ out,
                                                         int n_ctile, int nblk, int gt, float* __restrict__ stats,
                                                         int stat_cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile order: see gemm256.hip (XCD-contiguous id ranges, panels of gt token tiles x all channel tiles)
    const int id = xcd_remap(blockIdx.x, nblk);
    const int n_ttile = nblk / n_ctile;
    const int panel = id / (gt * n_ctile);
    const int gtp = min(gt, n_ttile - panel * gt);
    const int idp = id - panel * gt * n_ctile;
    const int ctile = idp / gtp, ttile = panel * gt + (idp - ctile * gtp);
    const int c0 = ctile * Cfg<NQM>::TC, t0 = ttile * TT;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4)
        gemm288_body<NQM, true, STATS>(A, lda, W, ldw, M, N, ktiles, alpha, out, c0, t0, smem, stats, stat_cols);
    else
        gemm288_body<NQM, false, STATS>(A, lda, W, ldw, M, N, ktiles, alpha, out, c0, t0, smem, stats, stat_cols);
}

template <int NQM, bool STATS = false>
static int launch288(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s,
                     float* stats = nullptr, int stat_cols = 0) {
    using C = Cfg<NQM>;
    const int Kpad = round_up(K, BK);
    const int n_ctile = N / C::TC;
    const int n_ttile = (M + TT - 1) / TT;
    const int nblk = n_ctile * n_ttile;
    int gt = esmk_switch("ESMK_GEMM_GT", 4);  // panel width of the tile order (4: best or within noise on every shape)
    if (gt < 1) gt = 1;
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm288_kernel<NQM, STATS>, C::LDS_BYTES, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", (int)(C::LDS_BYTES));
    hipLaunchKernelGGL((gemm288_kernel<NQM, STATS>), dim3(nblk), dim3(512), C::LDS_BYTES, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, alpha, (__bf16*)out, n_ctile, nblk, gt, stats, stat_cols);
    ESMK_CHECK_LAUNCH("esmk_gemm(288/192)");
    return ESMK_OK;
}

}  // namespace

// Called by esmk_gemm (gemm.hip): N % 288 == 0 (or N % 192 == 0 for esmk_gemm192), bf16 output.  W packed with rows
// padded to 256 (>= N rows exist).
int esmk_gemm288(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s) {
    return launch288<3>(A, lda, W, M, N, K, alpha, out, s);
}
int esmk_gemm192(const void* A, int lda, const void* W, int M, int N, int K, float alpha, void* out, hipStream_t s) {
    return launch288<2>(A, lda, W, M, N, K, alpha, out, s);
}
// QKV projection (N = 3 d) that also emits the q-side LayerNorm partial sums; tc = 288 or 192 (d % tc == 0).
int esmk_gemm288_stats(const void* A, int lda, const void* W, int M, int d, int K, void* out, float* stats, int tc,
                       hipStream_t s) {
    if (tc == 288) return launch288<3, true>(A, lda, W, M, 3 * d, K, 1.f, out, s, stats, d);
    return launch288<2, true>(A, lda, W, M, 3 * d, K, 1.f, out, s, stats, d);
}
