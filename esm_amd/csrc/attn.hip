// Varlen packed flash attention, head dim 64, non-causal, block-diagonal by segment (cu_seqlens).
//
// One workgroup = 4 waves = 128 query rows of one (segment, head); each wave owns 32 query rows.
// K/V tiles of 64 keys stream through a small LDS ring by LDS-DMA (buffer_load ... lds: segment base in a scalar
// descriptor, per-lane byte offsets, the tile offset in an SGPR), one raw s_barrier per tile; four workgroups per CU
// (128 VGPRs, 32 KiB of LDS) cover each other's waits.
// Scores are computed "swapped": S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, so the query row sits on the lane and its
// keys in the registers; the exponentiated tile converts in place (bf16) into the B operand of O^T += V^T . P^T - no
// LDS round trip for P.  V^T fragments come from the row-major V tile through ds_read_b64_tr_b16.
//
// The loop was VALU-issue bound (rocprofv3 PMC, profiles/r02_pmc_attn_before.json: the vector ALU issued 68 % of the
// kernel's cycles, the MFMA pipe was busy 31 %), so the softmax is arranged to issue as little as possible per tile:
//   * q was pre-multiplied by softmax_scale*log2(e) (esmk_qk_prep), so p = exp2(s) needs no multiply;
//   * NO running maximum in the common path: p = exp2(s) is taken straight off the score accumulators.  exp2 and the
//     fp32 / bf16 formats are scale-free, so p = exp2(s) loses no accuracy against exp2(s - max); what must not happen
//     is overflow or total underflow, and |s| stays far inside the fp32 exponent range for normalised q / k (QK-LayerNorm:
//     |s| <= 0.18 |q_h| |k_h|, a few tens).  The guard: after the last tile every row sum must lie in [2^-100, 2^100]
//     (an inf or NaN fails too).  A block where any row fails is redone from its first tile by a second loop with the
//     textbook running maximum and per-tile rescale - workgroup-uniform decision, correct for every finite input.
//     Gone from the common path: the per-tile row maximum (32 max), the subtraction (32), the O rescale (32 mul) and
//     the cross-half exchange; per tile and wave 32 exp + 32 add + 16 cvt_pk remain beside the 16 MFMAs.
//
// LDS images (128-byte rows of 64 bf16): 16-byte chunk index XOR-swizzled,
//   K: chunk ^ ((row>>1)&7)       -> conflict-free ds_read_b128 row fragments
//   V: chunk ^ (((row>>1)&1)<<2)  -> conflict-free ds_read_b64_tr_b16 (4 rows x 64 B per half-wave)
// applied on the DMA source address and on the read address.
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int QBLK = ESMK_ATTN_QBLK;  // 128 query rows per workgroup
constexpr int WAVES = QBLK / 32;       // one wave per 32 query rows
constexpr int RPW = 64 / WAVES;    // K (and V) tile rows each wave stages
constexpr int KVBLK = 64;
constexpr int TILE = KVBLK * 128;     // bytes per K (or V) tile
constexpr int SLOT = 2 * TILE;        // one ring slot: [K | V]
#ifndef ESMK_ATTN_NSLOT
#define ESMK_ATTN_NSLOT 2
#endif
#ifndef ESMK_ATTN_SPLITK
#define ESMK_ATTN_SPLITK 0
#endif
#ifndef ESMK_ATTN_MINWG
#define ESMK_ATTN_MINWG 4
#endif
#ifndef ESMK_ATTN_DESC
#define ESMK_ATTN_DESC 1
#endif
#ifndef ESMK_ATTN_WIDE
#define ESMK_ATTN_WIDE 1
#endif
// Measured in one process (tools/attn_ab.py; C2 / C5 / varlen): 2 slots + 4 workgroups per CU (128 VGPRs, 32 KiB) 98.0 /
// 637 / 287 us; 3 slots + 3 per CU (130 VGPRs, 48 KiB) 103.8 / 655 / 299 us; round-1 kernel 116.7 / 791 / 359 us.
constexpr int NSLOT = ESMK_ATTN_NSLOT;   // 2: tiles staged one ahead; 3: two ahead with a counted vmcnt, fewer workgroups per CU
constexpr int LEAD = NSLOT - 1;
constexpr float L_HI = 1.2676506e30f, L_LO = 7.8886091e-31f;  // 2^100, 2^-100: safe range of a row sum (see the guard)

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// QF instantiation (esmk_attn_varlen_qfused): q arrives RAW from the QKV projection and its QK-LayerNorm (statistics
// over all d_model columns, attention.py:31-33,51-52), rotary embedding (rotary.py:47-65) and softmax scale are applied
// to the query fragments in registers, so that q makes no extra trip through HBM.  The row statistics come as partial
// (sum, sum of squares) pairs from the projection's epilogue (esmk_gemm_qkv_stats).
struct QFuse {
    const float* mean_rstd;  // [M][2]: mean and 1/sqrt(var + eps) of the token's q row (esmk_k_prep)
    const float* gamma;      // q_ln.weight [d]
    const float* beta;       // q_ln.bias [d] or NULL
    const int* positions;    // [M]
    const float* rcos;       // [rope_len][32]
    const float* rsin;
    int rope_len;
    float q_scale;
};

template <bool QF>
__global__ __launch_bounds__(WAVES * 64, ESMK_ATTN_MINWG) void attn_varlen_kernel(const __bf16* __restrict__ qkv,
                                                            __bf16* __restrict__ ctx,
                                                            const int* __restrict__ cu_seqlens,
                                                            const int* __restrict__ qblk_map, int n_qblk, int d,
                                                            int H, int nblk, int pad_tail, QFuse qf_) {
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLOT + 16];  // ring + the "redo this block" flag
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = xcd_remap(blockIdx.x, nblk);
    const int head = id / n_qblk, qb = id - head * n_qblk;
    const int seg = qblk_map[2 * qb], q0 = qblk_map[2 * qb + 1];
    const int s0 = cu_seqlens[seg], s1 = cu_seqlens[seg + 1];
    const int len = s1 - s0;
    const size_t ld = (size_t)3 * d;
    const int l31 = lane & 31, hi = lane >> 5;
    // pad_tail (batches described on the device, esmk_segments_from_tokens): segment `seg` is the valid prefix of a
    // padded row and segment seg+1 its padding; the query blocks of the row span both, rows past `len` get zeros
    const int tail_end = pad_tail ? cu_seqlens[seg + 2] - s0 : len;
    if (q0 >= len) {  // whole block is padding (or lies beyond the row): block-uniform exit before any barrier
        const int qr = q0 + wave * 32 + l31;
        if (qr < tail_end) {
            u32x4* op = (u32x4*)(ctx + (size_t)(s0 + qr) * d + head * 64 + hi * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) op[i] = u32x4{0u, 0u, 0u, 0u};
        }
        return;
    }

    bf16x8 qf[4];  // Q fragments (B operand of S^T = K.Q^T), filled below once the first K/V tile is on its way
    int qpos = 0;  // QF: the query's rotary position, requested first (its table rows are the only dependent loads)
    if constexpr (QF) {
        int qr = q0 + wave * 32 + l31;
        qr = qr < len ? qr : len - 1;
        qpos = qf_.positions[s0 + qr];
    }

    // ---- staging: each wave moves 2 K pieces + 2 V pieces (8 rows x 128 B each) per tile.  Per-lane byte offsets
    //      from the segment's first K / V row are computed once; the tile offset travels in an SGPR.  The last tile
    //      has its own offsets with rows clamped to the segment (keys past the end are masked by index below, but
    //      foreign or NaN rows must never enter the V operand: 0 * NaN = NaN). ----
    const int nt = (len + KVBLK - 1) / KVBLK;
    const unsigned ldb = (unsigned)(ld * 2);
    const __bf16* kbase = qkv + (size_t)s0 * ld + d + head * 64;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)(kbase + d), 0, 0x7fffffff, 0x00020000);
#if ESMK_ATTN_DESC
    static_assert(RPW == 16 || RPW == 8, "4 or 8 waves per workgroup");
    // Per-lane offsets of piece 0 inside a tile (piece 1 = row + 8: the K swizzle flips chunk bit 2, the V swizzle is
    // unchanged; ldb is a multiple of 128, so the XOR stays inside the 128-byte row).  The tile's first row goes into
    // the descriptor base and the rows that exist into its size (scalar arithmetic only): rows past the segment end
    // are out of range and the DMA writes zeros for them (keys past the end are masked by index below, and a zero V
    // row times p = 0 is 0 - foreign or NaN rows never enter the V operand).
    unsigned offK0, offV0;
    {
        const int r = wave * RPW + (lane >> 3);
        offK0 = (unsigned)r * ldb + (unsigned)((lane & 7) ^ ((r >> 1) & 7)) * 16u;
        offV0 = (unsigned)r * ldb + (unsigned)((lane & 7) ^ (((r >> 1) & 1) << 2)) * 16u;
    }
    auto stage = [&](int slot_off, int kt) {
        char* base = smem + slot_off + wave * (RPW * 128);
        const int rows = min(len - kt * KVBLK, KVBLK);
        const int nrec = (rows - 1) * (int)ldb + 128;
        const __bf16* kt0 = kbase + (size_t)kt * KVBLK * ld;
        const __amdgpu_buffer_rsrc_t tK = __builtin_amdgcn_make_buffer_rsrc((void*)kt0, 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t tV = __builtin_amdgcn_make_buffer_rsrc((void*)(kt0 + d), 0, nrec, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(tK, LDS_PTR(base), 16, (int)offK0, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(tV, LDS_PTR(base + TILE), 16, (int)offV0, 0, 0, 0);
        if constexpr (RPW == 16) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tK, LDS_PTR(base + 1024), 16, (int)((offK0 ^ 64u) + 8u * ldb), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tV, LDS_PTR(base + TILE + 1024), 16, (int)(offV0 + 8u * ldb), 0, 0, 0);
        }
    };
#else
    const int srow = wave * 16 + (lane >> 3);  // this lane's row inside the tile for piece 0 (piece 1: + 8)
    const int pchunk = lane & 7;
    auto stage = [&](int slot_off, int kt) {
        char* base = smem + slot_off + wave * 2048;
        const int so = kt * (int)(KVBLK * ldb);
        // rows of the last tile are clamped to the segment (computed here, once per block, instead of held in registers)
        const int last_rows = (kt == nt - 1) ? len - (nt - 1) * KVBLK : KVBLK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 8 * i;
            const int rl = r < last_rows ? r : last_rows - 1;
            const unsigned kc = (unsigned)(pchunk ^ ((r >> 1) & 7)) * 16u;
            const unsigned vc = (unsigned)(pchunk ^ (((r >> 1) & 1) << 2)) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, LDS_PTR(base + i * 1024), 16, (int)((unsigned)rl * ldb + kc), so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, LDS_PTR(base + TILE + i * 1024), 16, (int)((unsigned)rl * ldb + vc), so, 0, 0);
        }
    };

#endif

    // K fragment (A operand) read offsets: row = 32mt + l31, logical chunk = 2ks + hi
    int koff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int r = mt * 32 + l31;
        koff[mt] = r * 128 + ((hi ^ ((r >> 1) & 7)) << 4);  // ks = 0; ks adds (2ks)<<4 via XOR (bits 5,6)
    }
    // V^T fragment (A operand of O^T += V^T.P^T) through ds_read_b64_tr_b16:
    //   16-lane group g = (lane>>4)&1 -> dh columns 32t + 16g + i ; lane i=(lane&15): row q=i>>2, col piece p=i&3
    //   rows = keys 16s + 4hi + q (+8 for the second read)
    int voff[2];  // for the two reads (+0, +8 keys) at s = 0, t = 0
    {
        const int i = lane & 15, g = (lane >> 4) & 1, q = i >> 2, p = i & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 4 * hi + q + 8 * e;
            const int ch = 2 * g + (p >> 1);
            voff[e] = r * 128 + ((ch ^ (((r >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float lsum = 0.f;    // this half-wave's part of the row sum

    // S^T = K . Q^T for the 64 keys of the tile at kt_
    auto scores = [&](const char* kt_, f32x16 (&s)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            bf16x8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const bf16x8*)(kt_ + (koff[mt] ^ (ks << 5)));
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[mt], 0, 0, 0);
            if (ESMK_ATTN_SPLITK && mt == 0) __builtin_amdgcn_sched_barrier(0);  // at most one half's K fragments live
        }
    };
    auto mask_tail = [&](f32x16 (&s)[2], int kleft) {  // keys beyond the segment (last tile only)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + crow(r, hi) >= kleft) s[mt][r] = -INFINITY;
    };

    // V^T fragments are read by inline asm: hipcc treats the ds_read_tr builtin as a reader of the LDS-DMA destination
    // and drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of the first one of every tile, which would
    // serialise the ring.  Form: "=v" loads, later a wait statement naming every destination "+v" and a
    // sched_barrier(0) before the consuming MFMAs (cdna_hip_programming.md 5.7 item 1 (ii), rule 18).
    // Two register sets alternate over the four 16-key steps, so the reads of step sp+1 fly under the MFMAs of sp.
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    unsigned va[2][2];  // [read e][dh tile t] byte offsets inside a V tile (key step 0)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 2; ++t) va[e][t] = (unsigned)(voff[e] ^ (t << 6));
    struct VFrag { u32x2 a0, a1, b0, b1; };  // t = 0: (a0, a1), t = 1: (b0, b1)
#define VREAD(F, BASE, SP)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c8\n\tds_read_b64_tr_b16 %1, %5 offset:%c8\n\t"       \
                 "ds_read_b64_tr_b16 %2, %6 offset:%c8\n\tds_read_b64_tr_b16 %3, %7 offset:%c8"            \
                 : "=v"(F.a0), "=v"(F.a1), "=v"(F.b0), "=v"(F.b1)                                        \
                 : "v"(BASE + va[0][0]), "v"(BASE + va[1][0]), "v"(BASE + va[0][1]), "v"(BASE + va[1][1]), \
                   "i"((SP) * 2048)                                                                      \
                 : "memory")
#define VWAIT(F, N)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F.a0), "+v"(F.a1), "+v"(F.b0), "+v"(F.b1)::"memory"); \
    __builtin_amdgcn_sched_barrier(0)
    // k-step SP covers keys 16sp..16sp+15 = registers 8(sp&1).. of s[sp>>1]
    // (keys +16sp -> +2048sp bytes: 16 rows; row bits >= 4 do not touch either swizzle)
#define PV_STEP(F, SP)                                                                                  \
    {                                                                                                   \
        bf16x8 pf;                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[(SP) >> 1][8 * ((SP) & 1) + j]; \
        const u32x4 v0 = {F.a0.x, F.a0.y, F.a1.x, F.a1.y}, v1 = {F.b0.x, F.b0.y, F.b1.x, F.b1.y};       \
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o[0], 0, 0, 0); \
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o[1], 0, 0, 0); \
    }
    // O^T += V^T . P^T for the tile whose first V read (step 0 -> fa) is already in flight
#define PV_ALL()                                   \
    __builtin_amdgcn_sched_barrier(0);             \
    VREAD(fb, vbase, 1);                           \
    VWAIT(fa, 4);                                  \
    PV_STEP(fa, 0)                                 \
    __builtin_amdgcn_sched_barrier(0);             \
    VREAD(fa, vbase, 2);                           \
    VWAIT(fb, 4);                                  \
    PV_STEP(fb, 1)                                 \
    __builtin_amdgcn_sched_barrier(0);             \
    VREAD(fb, vbase, 3);                           \
    VWAIT(fa, 4);                                  \
    PV_STEP(fa, 2)                                 \
    __builtin_amdgcn_sched_barrier(0);             \
    VWAIT(fb, 0);                                  \
    PV_STEP(fb, 3)

    // One 64-key tile of the common path.  MASKED (compile-time: as a run-time condition hipcc if-converts the mask
    // into ~90 compares and selects that every tile would execute): the segment's last tile when it holds keys past
    // the end, which get -inf scores.
    auto tile = [&](auto masked_tag, int kt, int cur) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* kt_ = smem + cur;
        const unsigned vbase = lds0 + (unsigned)cur + TILE;
        f32x16 s[2];
        scores(kt_, s);
        VFrag fa, fb;
        VREAD(fa, vbase, 0);
        if (MASKED) mask_tail(s, len - kt * KVBLK);
        float ps = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[mt][r]);
                s[mt][r] = p;
                ps += p;
            }
        lsum += ps;
        PV_ALL();
    };

    int* redo_flag = (int*)(smem + NSLOT * SLOT);
    if (tid == 0) *redo_flag = 0;  // ordered before any later write by the barriers of the loop
    const bool ragged = (len & (KVBLK - 1)) != 0;  // the last tile holds keys past the end
    stage(0, 0);
    if (LEAD > 1 && nt > 1) stage(SLOT, 1);

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = l31][dh = 16ks + 8hi + j] ----
    {
        int qr = q0 + wave * 32 + l31;
        qr = qr < len ? qr : len - 1;
        const __bf16* qp = qkv + (size_t)(s0 + qr) * ld + head * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
        if constexpr (QF) {
            const size_t tok = (size_t)(s0 + qr);
            const f32x2 mr = *(const f32x2*)(qf_.mean_rstd + tok * 2);
            const float mean = mr[0], rstd = mr[1] * qf_.q_scale;
            const int pos = qpos < qf_.rope_len ? qpos : qf_.rope_len - 1;
            const float* gp = qf_.gamma + head * 64 + hi * 8;
            const float* bp = qf_.beta ? qf_.beta + head * 64 + hi * 8 : nullptr;  // kernel argument: uniform
            const float* cp = qf_.rcos + (size_t)pos * 32 + hi * 8;
            const float* sn = qf_.rsin + (size_t)pos * 32 + hi * 8;
            // channels of this lane: 16 ks + 8 hi + j; the rotary partner of ks is ks ^ 2 (column +-32), angle 16 (ks&1) + 8 hi + j
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                float a[8], b[8];  // normalised columns below 32 (ks = kl) and their partners (ks = kl + 2)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int ks = kl + 2 * half;
                    const u32x4 raw = __builtin_bit_cast(u32x4, qf[ks]);
                    const f32x4 g0 = *(const f32x4*)(gp + ks * 16), g1 = *(const f32x4*)(gp + ks * 16 + 4);
                    f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
                    if (bp) {
                        b0 = *(const f32x4*)(bp + ks * 16) * qf_.q_scale;
                        b1 = *(const f32x4*)(bp + ks * 16 + 4) * qf_.q_scale;
                    }
                    float* dst = half ? b : a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = __uint_as_float(raw[e] << 16), x1 = __uint_as_float(raw[e] & 0xffff0000u);
                        const float ga = (e < 2 ? g0[2 * e] : g1[2 * e - 4]) * rstd, gb = (e < 2 ? g0[2 * e + 1] : g1[2 * e - 3]) * rstd;
                        dst[2 * e] = (x0 - mean) * ga + (e < 2 ? b0[2 * e] : b1[2 * e - 4]);
                        dst[2 * e + 1] = (x1 - mean) * gb + (e < 2 ? b0[2 * e + 1] : b1[2 * e - 3]);
                    }
                }
                const f32x4 c0 = *(const f32x4*)(cp + kl * 16), c1 = *(const f32x4*)(cp + kl * 16 + 4);
                const f32x4 s0_ = *(const f32x4*)(sn + kl * 16), s1_ = *(const f32x4*)(sn + kl * 16 + 4);
                u32x4 lo, hi_;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ca = e < 2 ? c0[2 * e] : c1[2 * e - 4], cb = e < 2 ? c0[2 * e + 1] : c1[2 * e - 3];
                    const float sa = e < 2 ? s0_[2 * e] : s1_[2 * e - 4], sb = e < 2 ? s0_[2 * e + 1] : s1_[2 * e - 3];
                    lo[e] = pack_bf16x2(a[2 * e] * ca - b[2 * e] * sa, a[2 * e + 1] * cb - b[2 * e + 1] * sb);
                    hi_[e] = pack_bf16x2(b[2 * e] * ca + a[2 * e] * sa, b[2 * e + 1] * cb + a[2 * e + 1] * sb);
                }
                qf[kl] = __builtin_bit_cast(bf16x8, lo);
                qf[kl + 2] = __builtin_bit_cast(bf16x8, hi_);
            }
        }
    }

    {
        int cur = 0;           // byte offset of the ring slot holding tile kt
        int nxt = LEAD * SLOT; // slot that receives tile kt + LEAD (afterwards: the one read a tile ago)
        int kt = 0;
        const int n_plain = ragged ? nt - 1 : nt;  // tiles [0, n_plain) hold no key past the end
        for (; kt < n_plain; ++kt) {
            // tile kt has landed (this wave's pieces); the pieces of later tiles may stay in flight
            if (kt + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (LEAD - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading the slot refilled next
            __builtin_amdgcn_sched_barrier(0);
            if (kt + LEAD < nt) stage(nxt, kt + LEAD);
            tile(std::false_type{}, kt, cur);
            nxt = cur;
            cur = cur + SLOT == NSLOT * SLOT ? 0 : cur + SLOT;
        }
        if (ragged) {  // the segment's last tile holds keys past the end
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tile(std::true_type{}, kt, cur);
        }
    }

    // ---- guarded loop: only when some row sum of the block left the safe range above (scores beyond about +-100 in
    //      log2 units, or a NaN / inf came in).  The whole block is redone with the running row maximum and a rescale
    //      per tile.  Workgroup-uniform (the loop has barriers): OR over the waves via LDS. ----
    {
        const float ltot = lsum + __shfl_xor(lsum, 32);
        const int lane_g = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const bool bad = !(ltot > L_LO && ltot < L_HI) && (q0 + wave * 32 + (lane_g & 31) < len);  // !(..): inf / NaN too
        if (__any(bad) && lane == 0) *redo_flag = 1;
        __syncthreads();  // also: every wave is done with the ring
        const bool redo = *redo_flag != 0;
        if (__builtin_expect(redo, 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
            lsum = 0.f;
            float m = -INFINITY;
            stage(0, 0);
            for (int kt = 0; kt < nt; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int cur = (kt & 1) * SLOT;
                if (kt + 1 < nt) stage(SLOT - cur, kt + 1);
                const char* kt_ = smem + cur;
                const unsigned vbase = lds0 + (unsigned)cur + TILE;
                f32x16 s[2];
                scores(kt_, s);
                VFrag fa, fb;
                VREAD(fa, vbase, 0);
                if (kt == nt - 1) mask_tail(s, len - kt * KVBLK);
                float mx = s[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float mnew = fmaxf(m, mx);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                m = mnew;
                float ps = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s[mt][r] - mnew);
                        s[mt][r] = p;
                        ps += p;
                    }
                lsum = lsum * alpha + ps;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                PV_ALL();
            }
        }
    }
#undef PV_ALL
#undef PV_STEP
#undef VWAIT
#undef VREAD

    // ---- epilogue: O^T[dh = 32t + crow(r,hi)][q = l31] / l.  The lane's position is re-derived here (v_mbcnt) rather
    //      than kept alive through the loop: the kernel sits exactly at the 128-VGPR budget of four workgroups per CU ----
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / lsum;
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int l31e = lane_e & 31, hie = lane_e >> 5;
    const int qr = q0 + wave * 32 + l31e;
    if (qr >= len && qr < tail_end) {
        u32x4* zp = (u32x4*)(ctx + (size_t)(s0 + qr) * d + head * 64 + hie * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) zp[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if (qr < len) {
        __bf16* op = ctx + (size_t)(s0 + qr) * d + head * 64;
#if ESMK_ATTN_WIDE
        // The row is split over the half-waves in 8-byte pieces (lane: dh 8g + 4hi ..+3).  A half exchange per dword
        // (v_permlane32_swap: the upper half of the first operand swaps with the lower half of the second) on the
        // piece pairs (g, g+1) leaves 16 contiguous bytes in every lane: dh 8g..8g+7 below, 8g+8..8g+15 above.
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const unsigned ax = pack_bf16x2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
                const unsigned ay = pack_bf16x2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                const unsigned bx = pack_bf16x2(o[t][4 * g + 4] * inv, o[t][4 * g + 5] * inv);
                const unsigned by = pack_bf16x2(o[t][4 * g + 6] * inv, o[t][4 * g + 7] * inv);
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                *(u32x4*)(op + 32 * t + 8 * g + 8 * hie) = u32x4{sx[0], sy[0], sx[1], sy[1]};
            }
#else
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk.x = pack_bf16x2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
                pk.y = pack_bf16x2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                *(u32x2*)(op + 32 * t + 8 * g + 4 * hie) = pk;
            }
#endif
    }
}

}  // namespace

extern "C" int esmk_attn_plan(const int32_t* cu, int n_segments, int32_t* map, int capacity) {
    if (!cu || n_segments < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_plan: bad arguments");
    int n = 0;
    for (int s = 0; s < n_segments; ++s) {
        const int len = cu[s + 1] - cu[s];
        if (len < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_plan: cu_seqlens must be non-decreasing");
        for (int q = 0; q < len; q += QBLK) {
            if (map) {
                if (n >= capacity) ESMK_FAIL(ESMK_ERR_WORKSPACE, "esmk_attn_plan: capacity %d too small", capacity);
                map[2 * n] = s;
                map[2 * n + 1] = q;
            }
            ++n;
        }
    }
    return n;
}

int esmk_attn_varlen_impl(const void* qkv, void* ctx, const int32_t* cu_seqlens, const int32_t* qblk_map, int n_qblk,
                          int M, int d, int H, int pad_tail, esmk_stream_t stream);

extern "C" int esmk_attn_varlen(const void* qkv, void* ctx, const int32_t* cu_seqlens, const int32_t* qblk_map,
                                int n_qblk, int M, int d, int H, esmk_stream_t stream) {
    return esmk_attn_varlen_impl(qkv, ctx, cu_seqlens, qblk_map, n_qblk, M, d, H, 0, stream);
}

// pad_tail = 1: the batch layout of esmk_segments_from_tokens (esmk_batch.pad_tail)
int esmk_attn_varlen_impl(const void* qkv, void* ctx, const int32_t* cu_seqlens, const int32_t* qblk_map, int n_qblk,
                          int M, int d, int H, int pad_tail, esmk_stream_t stream) {
    if (n_qblk == 0 || M == 0) return ESMK_OK;
    if (!qkv || !ctx || !cu_seqlens || !qblk_map) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_varlen: null pointer");
    if (d != H * 64) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_attn_varlen: head dim must be 64 (d=%d, heads=%d)", d, H);
    if (n_qblk < 0 || M < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_varlen: bad sizes");
    if (n_qblk == 0 || M == 0) return ESMK_OK;
    const long long nblk = (long long)n_qblk * H;
    if (nblk > 0x7fffffffLL) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_attn_varlen: grid too large");
    hipLaunchKernelGGL(attn_varlen_kernel<false>, dim3((unsigned)nblk), dim3(WAVES * 64), 0, (hipStream_t)stream,
                       (const __bf16*)qkv, (__bf16*)ctx, cu_seqlens, qblk_map, n_qblk, d, H, (int)nblk, pad_tail, QFuse{});
    ESMK_CHECK_LAUNCH("esmk_attn_varlen");
    return ESMK_OK;
}

// q raw (as the QKV projection wrote it), k prepared by esmk_k_prep (which also reduced q's statistics); pad_tail as in esmk_batch
extern "C" int esmk_attn_varlen_qfused(const void* qkv, void* ctx, const int32_t* cu_seqlens, const int32_t* qblk_map,
                                       int n_qblk, int M, int d, int H, int pad_tail, const float* q_mean_rstd,
                                       const float* gq, const float* bq, const int32_t* positions, const float* rcos,
                                       const float* rsin, int rope_len, float q_scale, esmk_stream_t stream) {
    if (n_qblk == 0 || M == 0) return ESMK_OK;
    if (!qkv || !ctx || !cu_seqlens || !qblk_map || !q_mean_rstd || !gq || !positions || !rcos || !rsin)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_varlen_qfused: null pointer");
    if (d != H * 64) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_attn_varlen_qfused: head dim must be 64 (d=%d, heads=%d)", d, H);
    if (n_qblk < 0 || M < 0 || rope_len <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_varlen_qfused: bad sizes");
    if ((uintptr_t)q_mean_rstd & 7) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_attn_varlen_qfused: q_mean_rstd must be 8-byte aligned");
    const long long nblk = (long long)n_qblk * H;
    if (nblk > 0x7fffffffLL) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_attn_varlen_qfused: grid too large");
    QFuse qf{q_mean_rstd, gq, bq, positions, rcos, rsin, rope_len, q_scale};
    hipLaunchKernelGGL(attn_varlen_kernel<true>, dim3((unsigned)nblk), dim3(WAVES * 64), 0, (hipStream_t)stream,
                       (const __bf16*)qkv, (__bf16*)ctx, cu_seqlens, qblk_map, n_qblk, d, H, (int)nblk, pad_tail ? 1 : 0, qf);
    ESMK_CHECK_LAUNCH("esmk_attn_varlen_qfused");
    return ESMK_OK;
}
