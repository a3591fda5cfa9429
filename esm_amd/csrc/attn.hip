// Varlen packed flash attention, head dim 64, non-causal, block-diagonal by segment (cu_seqlens).
//
// One workgroup = 4 waves = 128 query rows of one (segment, head); each wave owns 32 query rows.
// K/V tiles of 64 keys stream through double-buffered LDS (global_load_lds_dwordx4, one barrier per tile).
// Scores are computed "swapped": S^T = K . Q^T with v_mfma_f32_32x32x16_bf16, so the query row sits on
// the lane and its keys in the registers: the row max / row sum are register reductions plus ONE exchange
// with lane^32, and the exponentiated tile converts in place (bf16) into the B operand of
// O^T += V^T . P^T - no LDS round trip for P.  V^T fragments come from the row-major V tile through
// ds_read_b64_tr_b16.  q was pre-multiplied by softmax_scale*log2(e) (esmk_qk_prep) so p = exp2(s - m).
//
// LDS images (128-byte rows of 64 bf16): 16-byte chunk index XOR-swizzled,
//   K: chunk ^ ((row>>1)&7)       -> conflict-free ds_read_b128 row fragments
//   V: chunk ^ (((row>>1)&1)<<2)  -> conflict-free ds_read_b64_tr_b16 (4 rows x 64 B per half-wave)
// applied on the DMA source address and on the read address.
#include "common.hpp"

namespace {

constexpr int QBLK = ESMK_ATTN_QBLK;  // 128 query rows per workgroup
constexpr int KVBLK = 64;
constexpr int TILE = KVBLK * 128;  // bytes per K (or V) tile

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__global__ __launch_bounds__(256, 2) void attn_varlen_kernel(const __bf16* __restrict__ qkv,
                                                            __bf16* __restrict__ ctx,
                                                            const int* __restrict__ cu_seqlens,
                                                            const int* __restrict__ qblk_map, int n_qblk, int d,
                                                            int H, int nblk, int pad_tail) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];  // [buf][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = l31][dh = 16ks + 8hi + j] ----
    bf16x8 qf[4];
    {
        int qr = q0 + wave * 32 + l31;
        qr = qr < len ? qr : len - 1;
        const __bf16* qp = qkv + (size_t)(s0 + qr) * ld + head * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
    }

    // ---- staging: each wave moves 2 K pieces + 2 V pieces (8 rows x 128 B each) per tile ----
    const int prow = lane >> 3, pchunk = lane & 7;
    const __bf16* kbase = qkv + (size_t)s0 * ld + d + head * 64;
    const __bf16* vbase = qkv + (size_t)s0 * ld + 2 * d + head * 64;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave * 2 + i;
            const int r = p * 8 + prow;  // row inside the tile
            int kr = kt * KVBLK + r;
            kr = kr < len ? kr : len - 1;  // clamp: masked by index below, but never read foreign/NaN rows
            const int kc = pchunk ^ ((r >> 1) & 7);
            const int vc = pchunk ^ (((r >> 1) & 1) << 2);
            __builtin_amdgcn_global_load_lds(GLB_PTR(kbase + (size_t)kr * ld + kc * 8), LDS_PTR(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(vbase + (size_t)kr * ld + vc * 8), LDS_PTR(base + TILE + p * 1024), 16, 0, 0);
        }
    };

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
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    const int nt = (len + KVBLK - 1) / KVBLK;
    stage(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nt) stage((kt + 1) & 1, kt + 1);
        const char* kt_ = smem + (kt & 1) * 2 * TILE;
        const char* vt_ = kt_ + TILE;

        // ---- S^T = K . Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[mt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kt_ + (koff[mt] ^ (ks << 5)));
                s[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[mt], 0, 0, 0);
            }
        }
        // ---- mask keys beyond the segment (last tile only) ----
        const int kleft = len - kt * KVBLK;
        if (kleft < KVBLK) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mt * 32 + crow(r, hi) >= kleft) s[mt][r] = -INFINITY;
        }
        // ---- online softmax (row = lane&31, its keys are spread over 32 registers x 2 half-waves) ----
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(m, mx);  // finite: every tile holds at least one unmasked key
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

        // ---- O^T += V^T . P^T : k-step sp covers keys 16sp..16sp+15 = registers 8(sp&1).. of s[sp>>1] ----
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[sp >> 1][8 * (sp & 1) + j];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // keys +16sp -> +2048sp bytes (16 rows; row bits >= 4 do not touch either swizzle);
                // dh tile t -> logical chunk + 4t  (XOR with 4<<4 = 64)
                const char* vb = vt_ + sp * 2048;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + (voff[0] ^ (t << 6))));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + (voff[1] ^ (t << 6))));
                const s16x8 vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[t], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: O^T[dh = 32t + crow(r,hi)][q = l31] / l ----
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / lsum;
    const int qr = q0 + wave * 32 + l31;
    if (qr >= len && qr < tail_end) {
        u32x4* zp = (u32x4*)(ctx + (size_t)(s0 + qr) * d + head * 64 + hi * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) zp[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if (qr < len) {
        __bf16* op = ctx + (size_t)(s0 + qr) * d + head * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk;
                pk.x = pack_bf16x2(o[t][4 * g] * inv, o[t][4 * g + 1] * inv);
                pk.y = pack_bf16x2(o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                *(u32x2*)(op + 32 * t + 8 * g + 4 * hi) = pk;
            }
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
    hipLaunchKernelGGL(attn_varlen_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)qkv, (__bf16*)ctx, cu_seqlens, qblk_map, n_qblk, d, H, (int)nblk, pad_tail);
    ESMK_CHECK_LAUNCH("esmk_attn_varlen");
    return ESMK_OK;
}
