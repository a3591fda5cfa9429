// 256x256x64 bf16 MFMA GEMM with FOUR waves and 128x128 wave tiles (one wave per SIMD, 512-register budget:
// 256 accumulator registers + 96 fragment registers per lane).  Experimental sibling of gemm256.hip
// (ESMK_GEMM_4W=1): same orientation (MFMA M = output channel, MFMA N = token), same LDS image (128-B rows,
// 16-byte chunk XOR-swizzled by (row>>1)&7), same LDS-DMA pipeline 1.5 K-tiles ahead with counted vmcnt and two
// raw barriers per K-tile, same epilogues.  What changes is the work per wave: a 128x128 wave tile reads 256 B of
// LDS per MFMA instead of 384 B, and a barrier joins 4 waves instead of 8.
//
// Wave (wm, wn) = (wave>>1, wave&1) owns channels wm*128..+127 x tokens wn*128..+127.  A K-tile (BK = 64) is 128
// v_mfma_f32_16x16x32_bf16 per wave in four blocks (h, kk): h = 64-channel half of the wave tile, kk = 32-deep
// half of the K-tile; each block is 4 (mi) x 8 (ni) MFMAs.
//
// LDS: 2 K-tile buffers x 4 UNITS of 16 KiB (128 rows x 128 B):
//   U0 = W rows of h=0 (both wm), U1 = A rows of ni 0-3 (both wn), U2 = A rows of ni 4-7, U3 = W rows of h=1.
#include <stdlib.h>

#include "common.hpp"
#include "gemm_epi.hpp"

namespace {

constexpr int TB = 256;
constexpr int BK = 64;
constexpr int UNIT = 128 * 128;
constexpr int KBUF = 4 * UNIT;
constexpr int LDS_BYTES = 4 * 128 * (256 + 16);  // 136 KiB: max(K-loop buffers 128 KiB, epilogue staging)

#define SEG_BARRIER()                        \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

constexpr int EPI_SWIGLU_BIAS = 102;

template <int EPI_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(
    const __bf16* __restrict__ A, int lda, const __bf16* __restrict__ W, int ldw, int M, int N, int ktiles,
    const float* __restrict__ bias, float alpha, void* __restrict__ out, int n_ctile, int nblk, int gt) {
    constexpr bool kSwigluBias = (EPI_ == EPI_SWIGLU_BIAS);
    constexpr int EPI = kSwigluBias ? ESMK_EPI_SWIGLU : EPI_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // tile order: as gemm256.hip (XCD-contiguous ids, panels of gt token tiles x all channel tiles)
    const int id = xcd_remap(blockIdx.x, nblk);
    const int n_ttile = nblk / n_ctile;
    const int panel = id / (gt * n_ctile);
    const int gtp = min(gt, n_ttile - panel * gt);
    const int idp = id - panel * gt * n_ctile;
    const int ctile = idp / gtp, ttile = panel * gt + (idp - ctile * gtp);
    const int c0 = ctile * TB, t0 = ttile * TB;

    // ---- DMA sources: wave w moves pieces 4w..4w+3 (8 rows x 128 B each) of every unit ----
    const char* wbase = (const char*)(W + (size_t)c0 * ldw);
    const char* abase = (const char*)(A + (size_t)t0 * lda);
    unsigned soff[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);          // unit row 0..127
        const int lc = ((lane & 7) ^ ((r >> 1) & 7)) * 16;       // logical chunk byte offset in the 128-B row
        const int urow = (r >> 6) * 128 + (r & 63);              // tile row of the first unit of a pair (+64: second)
        int tr0 = urow, tr1 = urow + 64;
        tr0 = t0 + tr0 < M ? tr0 : M - 1 - t0;
        tr1 = t0 + tr1 < M ? tr1 : M - 1 - t0;
        soff[0][i] = (unsigned)urow * (unsigned)(ldw * 2) + lc;
        soff[1][i] = (unsigned)tr0 * (unsigned)(lda * 2) + lc;
        soff[2][i] = (unsigned)tr1 * (unsigned)(lda * 2) + lc;
        soff[3][i] = (unsigned)(urow + 64) * (unsigned)(ldw * 2) + lc;
    }
    const int dma_dst = wave * 4096;  // byte offset of this wave's four pieces inside a unit

    // ---- fragment read offsets inside a unit (kk = 0, mi / ni = 0) ----
    const int ra = wm * 64 + (lane & 15), rb = wn * 64 + (lane & 15);
    const int a_off = ra * 128 + (((lane >> 4) ^ ((ra >> 1) & 7)) << 4);
    const int b_off = rb * 128 + (((lane >> 4) ^ ((rb >> 1) & 7)) << 4);

    f32x4 acc[2][4][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[8][2];  // A sub-tile of the current h; B fragments of all 128 tokens; [..][kk]

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
    auto issue_piece = [&](int u, int i, int kt) {
        char* dst = smem + (kt & 1) * KBUF + u * UNIT + dma_dst + i * 1024;
        const int koff = kt * (BK * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds((u == 0 || u == 3) ? rW : rA, LDS_PTR(dst), 16, (int)soff[u][i], koff, 0, 0);
    };
    auto issue_unit = [&](int u, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(u, i, kt);
    };
    auto read_a_kk = [&](const char* unit, int kk) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi][kk] = *(const bf16x8*)(unit + ((a_off + mi * 2048) ^ (kk << 6)));
    };
    auto read_b_kk = [&](const char* buf, int kk) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
            bfr[ni][kk] = *(const bf16x8*)(buf + (1 + (ni >> 2)) * UNIT + ((b_off + (ni & 3) * 2048) ^ (kk << 6)));
    };

    // ---- prologue: all units of tile 0 and tile 1; wait for U0..U2(0); load the kk=0 fragments ----
    issue_unit(0, 0);
    issue_unit(1, 0);
    issue_unit(2, 0);
    issue_unit(3, 0);
    if (ktiles > 1) {
        issue_unit(0, 1);
        issue_unit(1, 1);
        issue_unit(2, 1);
        issue_unit(3, 1);
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    SEG_BARRIER();
    read_a_kk(smem + 0 * UNIT, 0);
    read_b_kk(smem, 0);

    // 32 MFMAs of block (H, KK) as 8 groups of 4 (one mi, four ni); S0..S7 are statements issued behind group 0..7.
    // the accumulators are pinned to the 256 AGPRs ("+a"): left to itself the register allocator keeps some of them
    // in VGPRs and shuffles ~150 registers between the two files per K-tile
#define MF(H, KK, MI, NI) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[H][MI][NI]) : "v"(af[MI][KK]), "v"(bfr[NI][KK]));
#define MFGRP(H, KK, MI, NB, S)                                                                    \
    MF(H, KK, MI, NB + 0) MF(H, KK, MI, NB + 1) MF(H, KK, MI, NB + 2) MF(H, KK, MI, NB + 3)        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    S;                                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA_BLOCK(H, KK, S0, S1, S2, S3, S4, S5, S6, S7)                                          \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    MFGRP(H, KK, 0, 0, S0) MFGRP(H, KK, 0, 4, S1) MFGRP(H, KK, 1, 0, S2) MFGRP(H, KK, 1, 4, S3)    \
    MFGRP(H, KK, 2, 0, S4) MFGRP(H, KK, 2, 4, S5) MFGRP(H, KK, 3, 0, S6) MFGRP(H, KK, 3, 4, S7)    \
    __builtin_amdgcn_s_setprio(0);
#define NOP_ ((void)0)
#define P2(C, U, I, KT) { if (C) { issue_piece(U, I, KT); issue_piece(U, I + 1, KT); } }
#define P1(C, U, I, KT) { if (C) issue_piece(U, I, KT); }

    // One K-tile = four blocks of 32 MFMAs; every block also issues the LDS reads of fragments that a LATER block
    // consumes, into registers the PREVIOUS block retired:
    //   block 1 (h0,kk0): reads kk1 of A(h0)+B                 then  vmcnt / lgkmcnt(0) / barrier P
    //   block 2 (h0,kk1): reads kk0 of A(h1); DMA U0-2(kt+2), two pieces per MFMA group
    //   block 3 (h1,kk0): reads kk1 of A(h1)                   then  vmcnt / lgkmcnt(0) / barrier Q
    //   block 4 (h1,kk1): reads kk0 of A(h0)+B of tile kt+1; DMA U3(kt+2)
    // Barrier P(kt): U3(kt) has landed for every wave, and every wave is done with slots U0-2 of this buffer.
    // Barrier Q(kt): U0-2(kt+1) have landed, and every wave is done with slot U3 of this buffer.
    // Issue order per wave is U0-2(t) [12 pieces], U3(t) [4], U0-2(t+1), U3(t+1), ...: "all but the two youngest
    // groups" = s_waitcnt vmcnt(16).  M1: tile kt+1 exists, M2: tile kt+2 exists.
#define KTILE(M1, M2)                                                                              \
    {                                                                                              \
        const char* cur = smem + (kt & 1) * KBUF;                                                  \
        const char* nxt = smem + ((kt + 1) & 1) * KBUF;                                            \
        MFMA_BLOCK(0, 0, { read_a_kk(cur + 0 * UNIT, 1); read_b_kk(cur, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        if (M1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                  \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(0, 1, { read_a_kk(cur + 3 * UNIT, 0); }, P2(M2, 0, 0, kt + 2), P2(M2, 0, 2, kt + 2), P2(M2, 1, 0, kt + 2), \
                   P2(M2, 1, 2, kt + 2), P2(M2, 2, 0, kt + 2), P2(M2, 2, 2, kt + 2), NOP_)                     \
        MFMA_BLOCK(1, 0, { read_a_kk(cur + 3 * UNIT, 1); }, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_, NOP_) \
        if (M2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                  \
        else if (M1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
        SEG_BARRIER();                                                                             \
        MFMA_BLOCK(1, 1, { if (M1) { read_a_kk(nxt + 0 * UNIT, 0); read_b_kk(nxt, 0); } }, NOP_,   \
                   P1(M2, 3, 0, kt + 2), P1(M2, 3, 1, kt + 2), P1(M2, 3, 2, kt + 2), P1(M2, 3, 3, kt + 2), NOP_, NOP_) \
    }

    int kt = 0;
    for (; kt < ktiles - 2; ++kt) KTILE(true, true)
    if (ktiles >= 2) {
        KTILE(true, false)
        ++kt;
    }
    KTILE(false, false)
#undef KTILE
#undef MFMA_BLOCK
#undef MFGRP
#undef MF
#undef NOP_
#undef P1
#undef P2

    // ---- epilogue ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the inline-asm MFMAs are opaque to the hazard recognizer
    // Everything the epilogue computes starts from values laundered HERE: otherwise its address arithmetic is hoisted
    // into the K loop, the register pressure there spills accumulators, and a spill store straight behind an
    // inline-asm MFMA reads its AGPRs before the result has landed (the Makefile rejects any scratch use in this file).
    // The accumulators are laundered too (LAUNDER4, four at a time right where they are consumed): the epilogue then
    // reads values defined behind the s_nops and cannot copy an accumulator out of its AGPRs straight behind the
    // (opaque) MFMA that finished it, nor pile all 256 of them up in VGPRs ahead of the epilogue's branches.
#define LAUNDER4(H, NI)                                                                            \
    asm volatile("" : "+a"(acc[H][0][NI]), "+a"(acc[H][1][NI]), "+a"(acc[H][2][NI]), "+a"(acc[H][3][NI]))
    int t0e = t0, c0e = c0, lane_e = lane;
    asm volatile("" : "+s"(t0e), "+s"(c0e), "+v"(lane_e));
#define t0 t0e
#define c0 c0e
#define lane lane_e
    const int q4 = (lane >> 4) * 4;
    constexpr bool kBf16Out = (EPI == ESMK_EPI_BF16 || EPI == ESMK_EPI_SWIGLU || EPI == ESMK_EPI_BIAS_GELU);
    const int OW = (EPI == ESMK_EPI_SWIGLU) ? (N >> 1) : N;  // output width
    // bf16 outputs always take the staged path: esmk_gemm4w sends widths that are not multiples of 8 (rows not 16-byte
    // aligned) to the eight-wave kernel.  ONE epilogue path per instantiation also keeps the register allocator from
    // splitting an accumulator's live range through VGPRs inside the K loop (seen with two paths: a v_accvgpr_read at
    // the loop head straight behind the opaque MFMAs of the previous iteration).
    if constexpr (kBf16Out) {
        // bf16 outputs: stage this wave's 128-token x CW-channel block through its own LDS region (no wave reads
        // fragments after barrier Q of the last tile, and every DMA was retired at its barrier P) and store whole
        // rows, 16 B per lane.  No inter-wave synchronisation needed.
        constexpr int CW = (EPI == ESMK_EPI_SWIGLU) ? 64 : 128;     // channels per wave after the epilogue op
        constexpr int PITCH = CW * 2 + 16;                          // row pitch in bytes (+16: bank spread)
        char* stg = smem + wave * (128 * PITCH);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                __builtin_amdgcn_sched_barrier(0);  // keep the accumulator reads of later iterations from piling up
                LAUNDER4(h, ni);
                const int trow = ni * 16 + (lane & 15);
                if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                    for (int mi = 0; mi < 4; mi += 2) {
                        f32x4 g = acc[h][mi][ni], u = acc[h][mi + 1][ni];
                        if constexpr (kSwigluBias) {
                            const int c = ((c0 + wm * 128) >> 1) + h * 32 + (mi >> 1) * 16 + q4;  // output channel
                            if (c < OW) {
                                g += *(const f32x4*)(bias + c);
                                u += *(const f32x4*)(bias + OW + c);
                            }
                        }
                        u32x2 v;
                        v.x = pack_bf16x2(silu(g[0]) * u[0], silu(g[1]) * u[1]);
                        v.y = pack_bf16x2(silu(g[2]) * u[2], silu(g[3]) * u[3]);
                        *(u32x2*)(stg + trow * PITCH + (h * 32 + (mi >> 1) * 16 + q4) * 2) = v;
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        f32x4 v = acc[h][mi][ni];
                        const int cl = h * 64 + mi * 16 + q4;
                        if constexpr (EPI == ESMK_EPI_BIAS_GELU) {
                            const int c = c0 + wm * 128 + cl;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r] + (c + r < N ? bias[c + r] : 0.f));
                        } else {
                            v *= alpha;
                        }
                        u32x2 pk;
                        pk.x = pack_bf16x2(v[0], v[1]);
                        pk.y = pack_bf16x2(v[2], v[3]);
                        *(u32x2*)(stg + trow * PITCH + cl * 2) = pk;
                    }
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        constexpr int LPR = CW / 8;         // lanes per row (16 B each)
        constexpr int RPI = 64 / LPR;       // rows per wave-instruction
        const int cbase = (EPI == ESMK_EPI_SWIGLU) ? ((c0 + wm * 128) >> 1) : (c0 + wm * 128);
        const int cc = cbase + (lane % LPR) * 8;
#pragma unroll 8
        for (int it = 0; it < 128 / RPI; ++it) {
            const int trow = it * RPI + lane / LPR;
            const int t = t0 + wn * 128 + trow;
            const u32x4 v = *(const u32x4*)(stg + trow * PITCH + (lane % LPR) * 16);
            if (t < M && cc < OW) *(u32x4*)((__bf16*)out + (size_t)t * OW + cc) = v;
        }
        return;
    } else {
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
        __builtin_amdgcn_sched_barrier(0);
        const int t = t0 + wn * 128 + ni * 16 + (lane & 15);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            LAUNDER4(h, ni);
            if (t >= M) continue;
            const int cb = c0 + wm * 128 + h * 64;
            if constexpr (EPI == ESMK_EPI_SWIGLU) {
#pragma unroll
                for (int mi = 0; mi < 4; mi += 2)
                    epi_swiglu4(acc[h][mi][ni], acc[h][mi + 1][ni], t, (cb >> 1) + (mi >> 1) * 16 + q4, N >> 1, out,
                                kSwigluBias ? bias : nullptr);
            } else {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) epi_store4<EPI>(acc[h][mi][ni], t, cb + mi * 16 + q4, N, bias, alpha, out);
            }
        }
    }
    }
}

#undef t0
#undef c0
#undef lane
#undef LAUNDER4

template <int EPI>
int launch(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, float alpha, void* out,
           hipStream_t s, int n_ctile) {
    const int Kpad = round_up(K, BK);
    const int n_ttile = (M + TB - 1) / TB;
    const int nblk = n_ctile * n_ttile;
    static int gt = 0;
    if (gt == 0) { gt = esmk_switch("ESMK_GEMM_GT", 4); if (gt < 1) gt = 1; }
    static std::atomic<unsigned long long> lds_ready{0};
    if (!esmk_reserve_lds((const void*)gemm4w_kernel<EPI>, LDS_BYTES, lds_ready))
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_gemm: cannot reserve %d bytes of LDS", (int)(LDS_BYTES));
    hipLaunchKernelGGL(gemm4w_kernel<EPI>, dim3(nblk), dim3(256), LDS_BYTES, s, (const __bf16*)A, lda,
                       (const __bf16*)W, Kpad, M, N, Kpad / BK, bias, alpha, out, n_ctile, nblk, gt);
    ESMK_CHECK_LAUNCH("esmk_gemm(4w)");
    return ESMK_OK;
}

}  // namespace

int esmk_gemm256(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                 float alpha, void* out, hipStream_t s, int n_ctile);

// Same contract as esmk_gemm256 (gemm256.hip), which also takes the bf16-output widths that are not multiples of 8.
int esmk_gemm4w(const void* A, int lda, const void* W, int M, int N, int K, int epilogue, const float* bias,
                float alpha, void* out, hipStream_t s, int n_ctile) {
    const bool bf16_out = epilogue == ESMK_EPI_BF16 || epilogue == ESMK_EPI_SWIGLU || epilogue == ESMK_EPI_BIAS_GELU;
    const int ow = epilogue == ESMK_EPI_SWIGLU ? (N >> 1) : N;
    if (bf16_out && (ow & 7)) return esmk_gemm256(A, lda, W, M, N, K, epilogue, bias, alpha, out, s, n_ctile);
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
