// Output side of the VQ-VAE structure decoder (the stack that turns ESM3's structure tokens
// back into coordinates).  The 30-block decoder stack itself runs on esmk_trunk_forward; this file holds what
// follows it (reference esm/models/vqvae.py:395-437):
//   esmk_dim6_frames       Dim6RotStructureHead after its projection (esm/layers/structure_proj.py:46-63):
//                          translation scale, two normalised vectors -> Gram-Schmidt frame -> backbone atoms
//   esmk_categorical_mean  CategoricalMixture.mean (vqvae.py:114-136): softmax-weighted mean of bin mid-points (pLDDT)
//   esmk_pairwise_pae      PairwisePredictionHead on all L x L pairs (vqvae.py:79-95) fused with
//                          compute_predicted_aligned_error and compute_tm (predicted_aligned_error.py:37-66).
// The reference materialises [B, L, L, 128] pair features, the hidden layer of the same size and [B, L, L, 224]
// logits in fp32 (L = 512: 134 + 134 + 235 MB) and reads them back three times; here a wave builds the features
// of 16 pairs in registers straight into MFMA B-operand layout, runs both small GEMMs on the matrix cores with the
// weights held in registers, and only pae [B, L, L] and ptm [B] are ever written.
#include <float.h>

#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dim6_frames_kernel(const float* __restrict__ proj, int ldp, float trans_scale,
                                                          float* __restrict__ affine, float* __restrict__ bb, int M) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    const float* p = proj + (size_t)r * ldp;
    float t[3], x[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t[i] = p[i] * trans_scale;
        x[i] = p[3 + i];
        y[i] = p[6 + i];
    }
    const float nx = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) + 1e-5f;
    const float ny = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]) + 1e-5f;
    float xa[3], pl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // Affine3D.from_graham_schmidt(neg_x_axis = x + trans, origin = trans, xy_plane = y + trans): same roundings
        const float negx = x[i] / nx + t[i], xyp = y[i] / ny + t[i];
        xa[i] = t[i] - negx;
        pl[i] = xyp - t[i];
    }
    const float d0 = sqrtf(xa[0] * xa[0] + xa[1] * xa[1] + xa[2] * xa[2] + 1e-10f);
    float e0[3], e1[3], e2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) e0[i] = xa[i] / d0;
    const float dot = e0[0] * pl[0] + e0[1] * pl[1] + e0[2] * pl[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) e1[i] = pl[i] - e0[i] * dot;
    const float d1 = sqrtf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2] + 1e-10f);
#pragma unroll
    for (int i = 0; i < 3; ++i) e1[i] = e1[i] / d1;
    e2[0] = e0[1] * e1[2] - e0[2] * e1[1];
    e2[1] = e0[2] * e1[0] - e0[0] * e1[2];
    e2[2] = e0[0] * e1[1] - e0[1] * e1[0];
    float* a = affine + (size_t)r * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // basis vectors are the COLUMNS of the rotation (affine3d.py:507)
        a[3 * i + 0] = e0[i];
        a[3 * i + 1] = e1[i];
        a[3 * i + 2] = e2[i];
        a[9 + i] = t[i];
    }
    // ideal backbone atoms N, CA, C in the local frame (esm/utils/constants/physics.py:1-5)
    const float BB[3][3] = {{0.5256f, 1.3612f, 0.0f}, {0.0f, 0.0f, 0.0f}, {-1.5251f, 0.0f, 0.0f}};
    float* o = bb + (size_t)r * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) o[3 * k + i] = (e0[i] * BB[k][0] + e1[i] * BB[k][1] + e2[i] * BB[k][2]) + t[i];
}

// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void categorical_mean_kernel(const float* __restrict__ logits, int R, int V, float lo,
                                                               float width, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* l = logits + (size_t)r * V;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, l[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float z = 0.f, s = 0.f;
    for (int v = lane; v < V; v += 64) {
        const float e = expf(l[v] - mx);
        z += e;
        s += e * (lo + ((float)v + 0.5f) * width);
    }
    z = wave_sum(z);
    s = wave_sum(s);
    if (lane == 0) out[r] = s / z;
}

// ------------------------------------------------------------------------------------------------------------
// One wave per (batch row b, first pair index a); it walks the second index c in blocks of 16 pairs.
//   features of pair (a, c):  [q_c * k_a (64) | q_c - k_a (64)]   (vqvae.py:83-90; q, k = halves of the down-projection)
//   GEMM1  hidden[128 x 16 pairs] = W1 . features       MFMA 16x16x32 bf16, A = W1 rows, B = features
//   GELU(erf), LayerNorm statistics over the 128 hidden channels of each pair (gamma / beta are folded into W2 / b2)
//   GEMM2  logits[64 bins x 16 pairs] = W2g . normalised + b2
//   softmax over the 64 bins -> pae = sum p * bin centre, tm = sum p / (1 + (centre / d0)^2)
// The rows of W1 are fed to GEMM1 in a permuted order so that its accumulators ARE the B operand of GEMM2: output
// rows 4g+j of M-block 2kk+hb hold hidden channel 32kk + 8g + 4hb + j, i.e. lane group g owns channels 8g..8g+7 of
// every 32-channel K-step.
__global__ __launch_bounds__(256) void pairwise_pae_kernel(const float* __restrict__ qk, const __bf16* __restrict__ w1,
                                                           const __bf16* __restrict__ w2, const float* __restrict__ b2,
                                                           const uint8_t* __restrict__ aa, int L, float step,
                                                           float* __restrict__ pae, float* __restrict__ ptm) {
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (a >= L) return;  // whole wave; the kernel has no workgroup barrier
    const uint8_t* aab = aa + (size_t)b * L;
    float cnt = 0.f;
    for (int c = lane; c < L; c += 64) cnt += aab[c] ? 1.f : 0.f;
    cnt = wave_sum(cnt);
    const bool aa_a = aab[a] != 0;
    // d0 of the TM score (predicted_aligned_error.py:56)
    const float d0 = 1.24f * cbrtf(fmaxf(cnt, 19.f) - 15.f) - 1.8f;

    bf16x8 w1f[8][4], w2f[4][4];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int f = 32 * (m >> 1) + 8 * (n >> 2) + 4 * (m & 1) + (n & 3);  // hidden channel fed as row n of block m
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w1f[m][kk] = *(const bf16x8*)(w1 + f * 128 + 32 * kk + 8 * g);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w2f[m][kk] = *(const bf16x8*)(w2 + (16 * m + n) * 128 + 32 * kk + 8 * g);
    float b2v[4][4], fd[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bin = 16 * m + 4 * g + j;
            b2v[m][j] = b2[bin];
            const float ctr = ((float)bin + 0.5f) * step, rr = ctr / d0;
            fd[m][j] = 1.0f / (1.0f + rr * rr);
        }
    float kv[2][8];
    {
        const float* kp = qk + ((size_t)b * L + a) * 128 + 64 + 8 * g;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 u0 = *(const f32x4*)(kp + 32 * h), u1 = *(const f32x4*)(kp + 32 * h + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { kv[h][j] = u0[j]; kv[h][4 + j] = u1[j]; }
        }
    }
    float rowsum = 0.f;
    float* pae_row = pae + ((size_t)b * L + a) * L;
    for (int c0 = 0; c0 < L; c0 += 16) {
        const int c = c0 + n, cc = c < L ? c : L - 1;
        const float* qp = qk + ((size_t)b * L + cc) * 128 + 8 * g;
        bf16x8 fb[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 u0 = *(const f32x4*)(qp + 32 * h), u1 = *(const f32x4*)(qp + 32 * h + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float q = j < 4 ? u0[j] : u1[j - 4];
                fb[h][j] = (__bf16)(q * kv[h][j]);
                fb[2 + h][j] = (__bf16)(q - kv[h][j]);
            }
        }
        f32x4 h1[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) h1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < 8; ++m) h1[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[m][kk], fb[kk], h1[m], 0, 0, 0);
        // GELU (erf form, nn.GELU default) and LayerNorm statistics of this lane's pair n (channels spread over g)
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = h1[m][j];
                const float ge = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                h1[m][j] = ge;
                s += ge;
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 128.0f);
        float v2 = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = h1[m][j] - mean;
                v2 += dlt * dlt;
            }
        v2 += __shfl_xor(v2, 16);
        v2 += __shfl_xor(v2, 32);
        const float rstd = rsqrtf(v2 * (1.0f / 128.0f) + 1e-5f);
        f32x4 lg[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) lg[m] = f32x4{b2v[m][0], b2v[m][1], b2v[m][2], b2v[m][3]};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 nb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nb[j] = (__bf16)((h1[2 * kk][j] - mean) * rstd);
                nb[4 + j] = (__bf16)((h1[2 * kk + 1][j] - mean) * rstd);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) lg[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[m][kk], nb, lg[m], 0, 0, 0);
        }
        // pairs outside the residue x residue square: every logit = finfo.min -> uniform softmax (:46, :60)
        const bool sq = aa_a && (aab[cc] != 0);
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lg[m][j] = sq ? lg[m][j] : 0.f;
                mx = fmaxf(mx, lg[m][j]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float z = 0.f, se = 0.f, st = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = expf(lg[m][j] - mx);
                z += e;
                se += e * (((float)(16 * m + 4 * g + j) + 0.5f) * step);
                st += e * fd[m][j];
            }
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        se += __shfl_xor(se, 16);
        se += __shfl_xor(se, 32);
        st += __shfl_xor(st, 16);
        st += __shfl_xor(st, 32);
        if (g == 0 && c < L) {
            pae_row[c] = se / z;
            if (sq) rowsum += st / z;
        }
    }
    rowsum = wave_sum(rowsum);
    if (lane == 0) {
        // masked_mean over c (eps 1e-10), then max over a (:64-66).  Values are >= 0, so the integer order of the
        // bit patterns is the float order and atomicMax gives the same result in any arrival order.
        const float rowmean = aa_a ? rowsum / (1e-10f + cnt) : 0.f;
        atomicMax((int*)(ptm + b), __float_as_int(rowmean));
    }
}

}  // namespace

extern "C" int esmk_dim6_frames(const float* proj, int ldp, float trans_scale, float* affine, float* bb, int M,
                                esmk_stream_t stream) {
    if (M == 0) return ESMK_OK;
    if (!proj || !affine || !bb) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_dim6_frames: null pointer");
    if (M < 0 || ldp < 9) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_dim6_frames: need M >= 0 and ldp >= 9 (M=%d ldp=%d)", M, ldp);
    hipLaunchKernelGGL(dim6_frames_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, proj, ldp,
                       trans_scale, affine, bb, M);
    ESMK_CHECK_LAUNCH("esmk_dim6_frames");
    return ESMK_OK;
}

extern "C" int esmk_categorical_mean(const float* logits, int R, int V, float lo, float hi, float* out,
                                     esmk_stream_t stream) {
    if (R == 0) return ESMK_OK;
    if (!logits || !out) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_categorical_mean: null pointer");
    if (R < 0 || V <= 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_categorical_mean: need R >= 0, V > 0 (R=%d V=%d)", R, V);
    hipLaunchKernelGGL(categorical_mean_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, R, V, lo,
                       (hi - lo) / (float)V, out);
    ESMK_CHECK_LAUNCH("esmk_categorical_mean");
    return ESMK_OK;
}

extern "C" int esmk_pairwise_pae(const float* qk, const void* w1_bf16, const void* w2_bf16, const float* b2,
                                 const uint8_t* aa_mask, int B, int L, float max_bin, float* pae, float* ptm,
                                 esmk_stream_t stream) {
    if (B == 0 || L == 0) return ESMK_OK;
    if (!qk || !w1_bf16 || !w2_bf16 || !b2 || !aa_mask || !pae || !ptm)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pairwise_pae: null pointer");
    if (B < 0 || L < 0 || B > 65535) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pairwise_pae: need 0 <= B <= 65535, L >= 0");
    if (((uintptr_t)qk & 15) || ((uintptr_t)w1_bf16 & 15) || ((uintptr_t)w2_bf16 & 15))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pairwise_pae: qk / w1 / w2 must be 16-byte aligned");
    if (!(max_bin > 0.f)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_pairwise_pae: max_bin must be positive");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ptm, 0, (size_t)B * sizeof(float), s) != hipSuccess)
        ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_pairwise_pae: hipMemsetAsync failed");
    // 64 bins: centres (i + 0.5) * step with step = max_bin / 62 (predicted_aligned_error.py:20-29)
    hipLaunchKernelGGL(pairwise_pae_kernel, dim3((L + 3) / 4, B), dim3(256), 0, s, qk, (const __bf16*)w1_bf16,
                       (const __bf16*)w2_bf16, b2, aa_mask, L, max_bin / 62.0f, pae, ptm);
    ESMK_CHECK_LAUNCH("esmk_pairwise_pae");
    return ESMK_OK;
}
