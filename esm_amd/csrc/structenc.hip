// Input side of the VQ-VAE structure encoder (coordinates -> ESM3 structure tokens;
// reference esm/models/vqvae.py:179-322).  Its two geometric blocks run on the shared GEMM / LayerNorm /
// geometric-attention kernels; this file holds what surrounds them:
//   esmk_knn_neighbourhoods  find_knn_edges + knn_graph (vqvae.py:265-289, esm/utils/misc.py:135-174) fused with the
//                            gathers of encode_local_structure (vqvae.py:224-248): per residue the 16 nearest
//                            residues by CA distance (sequence distance where coordinates are missing), their
//                            frames / masks / sequence ids and the clamped relative-position index
//   esmk_vq_quantize         query-node selection + frameless zeroing + pre_vq_proj + EMACodebook.forward
//                            (vqvae.py:260-261,317-320; esm/layers/codebook.py:57-75): nearest of 4096 codes, in
//                            fp32 throughout so that the discrete result does not depend on bf16 rounding
#include <float.h>

#include "common.hpp"

namespace {

constexpr float MAX_SUPPORTED_DISTANCE = 1e6f;  // esm/utils/structure/affine3d.py (knn_graph's fall-back offset)

// (value, index) lexicographic minimum across the wave: ascending sort order with the lower index first on ties
__device__ __forceinline__ void wave_lexmin(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// one wave per query residue; 4 queries per block; distances of the query to all L residues live in LDS
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ coords, const uint8_t* __restrict__ cmask,
                                                  const uint8_t* __restrict__ pad, const int* __restrict__ seq_id,
                                                  const int64_t* __restrict__ res_index,
                                                  const float* __restrict__ affine, int L, int E, int rel_bins,
                                                  int64_t* __restrict__ edges, uint8_t* __restrict__ edge_mask,
                                                  float* __restrict__ knn_affine, uint8_t* __restrict__ knn_mask,
                                                  int* __restrict__ knn_seq, int64_t* __restrict__ rel_idx) {
    extern __shared__ float dist_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w, b = blockIdx.y;
    if (i >= L) return;  // whole wave; no workgroup barrier below
    float* dist = dist_all + (size_t)w * L;
    const size_t base = (size_t)b * L;
    const bool ci = cmask[base + i] != 0, pi = pad[base + i] != 0;
    const int si = seq_id ? seq_id[base + i] : 0;
    const float* cai = coords + (base + i) * 9 + 3;
    const float xi = ci ? cai[0] : 0.f, yi = ci ? cai[1] : 0.f, zi = ci ? cai[2] : 0.f;
    for (int j = lane; j < L; j += 64) {
        const bool cj = cmask[base + j] != 0;
        float d;
        if (ci && cj) {
            const float* caj = coords + (base + j) * 9 + 3;
            const float dx = xi - caj[0], dy = yi - caj[1], dz = zi - caj[2];
            d = sqrtf(dx * dx + dy * dy + dz * dz);
        } else {
            d = (float)abs(i - j) * 1e2f + MAX_SUPPORTED_DISTANCE;  // misc.py:162-166
        }
        if (pi || pad[base + j] || (seq_id && seq_id[base + j] != si)) d = INFINITY;
        dist[j] = d;
    }
    // E rounds of "smallest (distance, index) strictly after the previous pick" = the first E entries of a stable
    // ascending sort (torch.sort on the CPU is stable; ties: equal sequence distances on both sides, +inf)
    float pv = -INFINITY;
    int pj = -1;
    int64_t res0 = 0;
    for (int e = 0; e < E; ++e) {
        float bv = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lane; j < L; j += 64) {
            const float d = dist[j];
            const bool after = d > pv || (d == pv && j > pj);
            if (after && (d < bv || (d == bv && j < bj))) { bv = d; bj = j; }
        }
        wave_lexmin(bv, bj);  // E <= L guarantees a candidate
        pv = bv;
        pj = bj;
        const size_t o = (base + i) * E + e;
        const int64_t rj = res_index ? res_index[base + bj] : (int64_t)bj;
        if (e == 0) res0 = rj;
        if (lane == 0) {
            edges[o] = bj;
            edge_mask[o] = isfinite(bv) ? 1 : 0;
            knn_mask[o] = cmask[base + bj];
            knn_seq[o] = seq_id ? seq_id[base + bj] : 0;
            int64_t dlt = rj - res0;  // RelativePositionEmbedding (vqvae.py:45-48)
            dlt = dlt < -rel_bins ? -rel_bins : (dlt > rel_bins ? rel_bins : dlt);
            rel_idx[o] = dlt + rel_bins + 1;
        }
        if (lane < 12) knn_affine[o * 12 + lane] = affine[(base + bj) * 12 + lane];
    }
}

// ------------------------------------------------------------------------------------------------------------
// 16 tokens per block.  x rows are taken with a stride (the query node is row 0 of each neighbourhood).
// fp32 FMAs on the vector ALUs; every LDS read is a 16-byte broadcast that feeds 16 (projection) or 64 (code search)
// FMAs, so one wave per SIMD (the tile takes 100 KB of LDS) still keeps its ALU busy.
constexpr int VQ_TOK = 16;

__global__ __launch_bounds__(256) void vq_quantize_kernel(const float* __restrict__ x, size_t row_stride,
                                                          const uint8_t* __restrict__ keep,
                                                          const float* __restrict__ wT /*[d][dz]*/,
                                                          const float* __restrict__ bias /*[dz]*/,
                                                          const float* __restrict__ codesT /*[dz][n_codes]*/,
                                                          const float* __restrict__ code_sq /*[n_codes]*/, int M, int d,
                                                          int n_codes, float* __restrict__ z_out,
                                                          float* __restrict__ zq_out, int64_t* __restrict__ idx_out) {
    constexpr int DZ = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                        // [VQ_TOK][d]
    float* zsT = smem + (size_t)VQ_TOK * d;  // [DZ][VQ_TOK]
    float* part = zsT + DZ * VQ_TOK;         // [VQ_TOK][DZ] partial sums of the upper k half; later bestv [VQ_TOK][256]
    float* bestv = part;
    int* besti = (int*)(bestv + VQ_TOK * 256);
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * VQ_TOK;
    for (int e = tid; e < VQ_TOK * d; e += 256) {
        const int t = e / d, k = e - t * d;
        const int tok = t0 + t;
        float v = 0.f;
        if (tok < M && (!keep || keep[tok])) v = x[(size_t)tok * row_stride + k];  // masked_fill(~affine_mask, 0)
        xs[e] = v;
    }
    __syncthreads();
    {   // z = x . W^T + b : thread = (channel c, half of k), all 16 tokens; d % 8 == 0
        const int c = tid & (DZ - 1), kh = tid >> 7;
        const int k0 = kh * (d >> 1), k1 = k0 + (d >> 1);
        float acc[VQ_TOK];
#pragma unroll
        for (int t = 0; t < VQ_TOK; ++t) acc[t] = 0.f;
#pragma unroll 4
        for (int k = k0; k < k1; k += 4) {
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = wT[(size_t)(k + u) * DZ + c];
#pragma unroll
            for (int t = 0; t < VQ_TOK; ++t) {
                const f32x4 xv = *(const f32x4*)(xs + t * d + k);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t] = fmaf(xv[u], wv[u], acc[t]);
            }
        }
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < VQ_TOK; ++t) part[t * DZ + c] = acc[t];
        }
        __syncthreads();
        if (kh == 0) {
            const float bv = bias[c];
#pragma unroll
            for (int t = 0; t < VQ_TOK; ++t) {
                const float z = (acc[t] + part[t * DZ + c]) + bv;
                zsT[c * VQ_TOK + t] = z;
                if (t0 + t < M && z_out) z_out[(size_t)(t0 + t) * DZ + c] = z;
            }
        }
    }
    __syncthreads();
    // distance up to the per-token constant |z|^2:  |e|^2 - 2 z.e   (codebook.py:63-67), argmin with first-index ties
    float bv[VQ_TOK];
    int bi[VQ_TOK];
#pragma unroll
    for (int t = 0; t < VQ_TOK; ++t) { bv[t] = INFINITY; bi[t] = 0x7fffffff; }
    for (int code0 = tid; code0 < n_codes; code0 += 1024) {  // this thread: codes code0 + 256 u, ascending
        float acc[4][VQ_TOK];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < VQ_TOK; ++t) acc[u][t] = 0.f;
#pragma unroll 8  // 32 independent global loads in flight per wave: the loop is load-latency bound otherwise
        for (int c = 0; c < DZ; ++c) {
            float ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int code = code0 + 256 * u;
                ev[u] = code < n_codes ? codesT[(size_t)c * n_codes + code] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 zv = *(const f32x4*)(zsT + c * VQ_TOK + 4 * q);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[u][4 * q + j] = fmaf(zv[j], ev[u], acc[u][4 * q + j]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int code = code0 + 256 * u;
            if (code < n_codes) {
                const float sq = code_sq[code];
#pragma unroll
                for (int t = 0; t < VQ_TOK; ++t) {
                    const float dd = sq - 2.0f * acc[u][t];
                    if (dd < bv[t]) { bv[t] = dd; bi[t] = code; }
                }
            }
        }
    }
    __syncthreads();  // part / bestv share LDS
#pragma unroll
    for (int t = 0; t < VQ_TOK; ++t) { bestv[t * 256 + tid] = bv[t]; besti[t * 256 + tid] = bi[t]; }
    __syncthreads();
    // one wave reduces 4 tokens
    const int lane = tid & 63, w = tid >> 6;
    for (int t = w * 4; t < w * 4 + 4; ++t) {
        float v = INFINITY;
        int ix = 0x7fffffff;
        for (int e = lane; e < 256; e += 64) {
            const float ov = bestv[t * 256 + e];
            const int oi = besti[t * 256 + e];
            if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        wave_lexmin(v, ix);
        const int tok = t0 + t;
        if (tok < M) {
            if (lane == 0) idx_out[tok] = ix;
            if (zq_out)
                for (int c = lane; c < DZ; c += 64) zq_out[(size_t)tok * DZ + c] = codesT[(size_t)c * n_codes + ix];
        }
    }
}

}  // namespace

extern "C" int esmk_knn_neighbourhoods(const float* coords, const uint8_t* coord_mask, const uint8_t* padding_mask,
                                       const int32_t* seq_id, const int64_t* residue_index, const float* affine, int B,
                                       int L, int knn, int rel_bins, int64_t* edges, uint8_t* edge_mask,
                                       float* knn_affine, uint8_t* knn_mask, int32_t* knn_seq, int64_t* rel_idx,
                                       esmk_stream_t stream) {
    if (B == 0 || L == 0) return ESMK_OK;
    if (!coords || !coord_mask || !padding_mask || !affine || !edges || !edge_mask || !knn_affine || !knn_mask || !knn_seq ||
        !rel_idx)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_knn_neighbourhoods: null pointer");
    if (B < 0 || L < 0 || B > 65535 || knn <= 0 || rel_bins < 0)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_knn_neighbourhoods: bad sizes B=%d L=%d knn=%d", B, L, knn);
    if (knn > L) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_knn_neighbourhoods: knn=%d > L=%d (pass min(knn, L) as the reference does)", knn, L);
    const size_t smem = (size_t)4 * L * sizeof(float);
    if (smem > 160 * 1024) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_knn_neighbourhoods: L=%d > 10240 residues per row", L);
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)knn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(knn_kernel, dim3((L + 3) / 4, B), dim3(256), smem, (hipStream_t)stream, coords, coord_mask,
                       padding_mask, seq_id, residue_index, affine, L, knn, rel_bins, edges, edge_mask, knn_affine,
                       knn_mask, knn_seq, rel_idx);
    ESMK_CHECK_LAUNCH("esmk_knn_neighbourhoods");
    return ESMK_OK;
}

extern "C" int esmk_vq_quantize(const float* x, int row_stride, const uint8_t* keep, const float* w_t, const float* bias,
                                const float* codes_t, const float* code_sq, int M, int d, int d_code, int n_codes,
                                float* z, float* z_q, int64_t* indices, esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!x || !w_t || !bias || !codes_t || !code_sq || !indices) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_vq_quantize: null pointer");
    if (M < 0 || d <= 0 || (d & 7) || n_codes <= 0 || row_stride < d || (row_stride & 3) || ((uintptr_t)x & 15))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_vq_quantize: need d %% 8 == 0, row_stride >= d and %% 4 == 0, x 16-byte aligned");
    if (d_code != 128) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_vq_quantize: code dimension must be 128 (got %d)", d_code);
    const size_t smem = ((size_t)VQ_TOK * d + VQ_TOK * 128 + 2 * VQ_TOK * 256) * sizeof(float);
    if (smem > 160 * 1024) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_vq_quantize: d=%d too wide for the LDS tile", d);
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)vq_quantize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(vq_quantize_kernel, dim3((M + VQ_TOK - 1) / VQ_TOK), dim3(256), smem, (hipStream_t)stream, x,
                       (size_t)row_stride, keep, w_t, bias, codes_t, code_sq, M, d, n_codes, z, z_q, indices);
    ESMK_CHECK_LAUNCH("esmk_vq_quantize");
    return ESMK_OK;
}
