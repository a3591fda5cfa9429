// ESM3 geometric attention (block 0) and backbone frames.  All fp32: the reference forces fp32 for
// rotations (affine3d.py:93,169) and the contraction length is 3, so this is VALU / exp / sqrt work, not MFMA.
#include <float.h>

#include "common.hpp"

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Gram-Schmidt frame from (N, CA, C): e0 = (CA-C)/|.|, e1 = orth(N-CA)/|.|, e2 = e0 x e1; R columns = e0,e1,e2.
__device__ __forceinline__ void gram_schmidt(V3 n, V3 ca, V3 c, float* aff) {
    const float eps = 1e-10f;  // Affine3D.from_graham_schmidt default (affine3d.py:417-429)
    V3 xa = sub(ca, c);
    V3 xy = sub(n, ca);
    V3 e0 = scale(xa, 1.0f / sqrtf(dot(xa, xa) + eps));
    V3 e1 = sub(xy, scale(e0, dot(e0, xy)));
    e1 = scale(e1, 1.0f / sqrtf(dot(e1, e1) + eps));
    V3 e2 = cross(e0, e1);
    aff[0] = e0.x; aff[1] = e1.x; aff[2] = e2.x;
    aff[3] = e0.y; aff[4] = e1.y; aff[5] = e2.y;
    aff[6] = e0.z; aff[7] = e1.z; aff[8] = e2.z;
    aff[9] = ca.x; aff[10] = ca.y; aff[11] = ca.z;
}

__device__ __forceinline__ bool coord_ok(float v) { return isfinite(v) && v < 1e6f; }

// one 256-thread block per batch row
__global__ __launch_bounds__(256) void frames_kernel(const float* __restrict__ coords, float* __restrict__ affine,
                                                     uint8_t* __restrict__ mask, int L) {
    __shared__ float red[4][10];
    __shared__ float bh[12];
    const int b = blockIdx.x;
    const float* cb = coords + (size_t)b * L * 9;
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    for (int i = threadIdx.x; i < L; i += 256) {
        float c[9];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            c[k] = cb[(size_t)i * 9 + k];
            ok = ok && coord_ok(c[k]);
        }
        if (ok) {
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += c[k];
            acc[9] += 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[10];
        for (int k = 0; k < 10; ++k) s[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        const float inv = 1.0f / (s[9] + 1e-8f);
        V3 n{s[0] * inv, s[1] * inv, s[2] * inv}, ca{s[3] * inv, s[4] * inv, s[5] * inv}, c{s[6] * inv, s[7] * inv, s[8] * inv};
        float a[12];
        gram_schmidt(n, ca, c, a);
        if (s[9] == 0.f) {  // no coordinates in this row: identity rotation (affine3d.py:547-552)
            a[0] = 1; a[1] = 0; a[2] = 0; a[3] = 0; a[4] = 1; a[5] = 0; a[6] = 0; a[7] = 0; a[8] = 1;
        }
        for (int k = 0; k < 12; ++k) bh[k] = a[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += 256) {
        float c[9];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            c[k] = cb[(size_t)i * 9 + k];
            ok = ok && coord_ok(c[k]);
        }
        float a[12];
        if (ok) {
            gram_schmidt(V3{c[0], c[1], c[2]}, V3{c[3], c[4], c[5]}, V3{c[6], c[7], c[8]}, a);
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) a[k] = bh[k];
        }
        float* o = affine + ((size_t)b * L + i) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = a[k];
        mask[(size_t)b * L + i] = ok ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Geometric attention.
// prep: per (token, head) rotate (and translate) the projected 3-vectors into the global frame.
//   Q record  [H][M][8]  = { qr*cr (3), qd*cd (3), -, - }
//   K record  [H][M][12] = { kr (3), kd*cd (3), v (3), frame_ok, seq_id bits, chain_id bits }
//   with cr = softplus(w_r[h])/sqrt(3)*log2e, cd = softplus(w_d[h])/sqrt(3)*log2e, so that
//   log2e * (rot*w_r - dist*w_d) = dot(qr', kr) - |qd' - kd'|.
// ------------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ V3 rot_apply(const float* R, V3 p) {
    return {R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z,
            R[6] * p.x + R[7] * p.y + R[8] * p.z};
}

__global__ __launch_bounds__(256) void geom_prep_kernel(const float* __restrict__ proj,
                                                        const float* __restrict__ affine,
                                                        const uint8_t* __restrict__ amask,
                                                        const int* __restrict__ seq_id,
                                                        const int* __restrict__ chain_id,
                                                        const float* __restrict__ dist_scale,
                                                        const float* __restrict__ rot_scale, float* __restrict__ qrec,
                                                        float* __restrict__ krec, int M, int H) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * H) return;
    const int t = (int)(idx / H), h = (int)(idx - (size_t)t * H);
    const float* A = affine + (size_t)t * 12;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = A[k];
    const V3 T{A[9], A[10], A[11]};
    const float* p = proj + (size_t)t * 15 * H;
    auto vec = [&](int v) { return V3{p[v * 3], p[v * 3 + 1], p[v * 3 + 2]}; };
    const float cr = softplus(rot_scale[h]) * (LOG2E * 0.57735026918962576f);
    const float cd = softplus(dist_scale[h]) * (LOG2E * 0.57735026918962576f);
    V3 qr = rot_apply(R, vec(h));
    V3 kr = rot_apply(R, vec(H + h));
    V3 vv = rot_apply(R, vec(2 * H + h));
    V3 qd = rot_apply(R, vec(3 * H + h));
    V3 kd = rot_apply(R, vec(4 * H + h));
    qd = {qd.x + T.x, qd.y + T.y, qd.z + T.z};
    kd = {kd.x + T.x, kd.y + T.y, kd.z + T.z};
    float* q = qrec + ((size_t)h * M + t) * 8;
    *(f32x4*)q = f32x4{qr.x * cr, qr.y * cr, qr.z * cr, qd.x * cd};
    *(f32x4*)(q + 4) = f32x4{qd.y * cd, qd.z * cd, 0.f, 0.f};
    float* k = krec + ((size_t)h * M + t) * 12;
    *(f32x4*)k = f32x4{kr.x, kr.y, kr.z, kd.x * cd};
    *(f32x4*)(k + 4) = f32x4{kd.y * cd, kd.z * cd, vv.x, vv.y};
    *(f32x4*)(k + 8) = f32x4{vv.z, amask[t] ? 1.f : 0.f, __int_as_float(seq_id[t]), __int_as_float(chain_id[t])};
}

constexpr int GQ = 128;  // queries per block (one per thread)
constexpr int GK = 128;  // keys per LDS tile

// ZF = mask_and_zero_frameless (geom_attention.py:145-146): true -> rows without a frame give zeros (ESM3);
// false (VQ-VAE structure encoder) -> they attend like any other query, and a query whose keys are ALL masked gets
// the reference's uniform softmax over every key of its row (all scores collapse to finfo.min).
template <bool ZF>
__global__ __launch_bounds__(GQ) void geom_attn_kernel(const float* __restrict__ qrec, const float* __restrict__ krec,
                                                       const float* __restrict__ affine,
                                                       const uint8_t* __restrict__ amask,
                                                       const int* __restrict__ seq_id,
                                                       const int* __restrict__ chain_id,
                                                       const int* __restrict__ row_cu, __bf16* __restrict__ out,
                                                       int ld_out, int M, int H) {
    __shared__ __attribute__((aligned(16))) float ktile[GK * 12];
    const int row = blockIdx.y, h = blockIdx.z;
    const int r0 = row_cu[row], r1 = row_cu[row + 1];
    const int q0 = r0 + blockIdx.x * GQ;
    if (q0 >= r1) return;  // block-uniform
    const int qi = q0 + threadIdx.x;
    const bool active = qi < r1;
    const int qt = active ? qi : r1 - 1;
    const float* q = qrec + ((size_t)h * M + qt) * 8;
    const f32x4 qa = *(const f32x4*)q, qb = *(const f32x4*)(q + 4);
    const int my_seq = seq_id[qt], my_chain = chain_id[qt];
    const bool framed = amask[qt] != 0;

    float m = -FLT_MAX, l = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
    float ux = 0.f, uy = 0.f, uz = 0.f;  // !ZF: plain sum of v over the row (uniform fallback)
    for (int k0 = r0; k0 < r1; k0 += GK) {
        const int nk = min(GK, r1 - k0);
        __syncthreads();
        const f32x4* src = (const f32x4*)(krec + ((size_t)h * M + k0) * 12);
        for (int i = threadIdx.x; i < nk * 3; i += GQ) ((f32x4*)ktile)[i] = src[i];
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            const f32x4 ka = *(const f32x4*)(ktile + j * 12);
            const f32x4 kb = *(const f32x4*)(ktile + j * 12 + 4);
            const f32x4 kc = *(const f32x4*)(ktile + j * 12 + 8);
            // masked keys (no frame, or another chain) carry finfo.min in the reference; a framed query always
            // has itself as an unmasked key, so their softmax weight is exactly 0 -> skip them.
            const bool ok = (kc[1] != 0.f) && (__float_as_int(kc[3]) == my_chain);
            const float dx = qa[3] - ka[3], dy = qb[0] - kb[0], dz = qb[1] - kb[1];
            float s = qa[0] * ka[0] + qa[1] * ka[1] + qa[2] * ka[2] - sqrtf(dx * dx + dy * dy + dz * dz);
            s += (__float_as_int(kc[2]) == my_seq) ? LOG2E : 0.f;
            if constexpr (!ZF) { ux += kb[2]; uy += kb[3]; uz += kc[0]; }
            if (ok) {
                if (s > m) {
                    const float a = __builtin_amdgcn_exp2f(m - s);
                    l *= a; ox *= a; oy *= a; oz *= a;
                    m = s;
                }
                const float p = __builtin_amdgcn_exp2f(s - m);
                l += p;
                ox += p * kb[2];
                oy += p * kb[3];
                oz += p * kc[0];
            }
        }
    }
    if (!active) return;
    __bf16* o = out + (size_t)qi * ld_out + h * 3;
    if constexpr (ZF) {
        if (!framed || l == 0.f) {  // mask_and_zero_frameless (geom_attention.py:145-146)
            o[0] = (__bf16)0.f; o[1] = (__bf16)0.f; o[2] = (__bf16)0.f;
            return;
        }
    } else if (l == 0.f) {
        l = (float)(r1 - r0); ox = ux; oy = uy; oz = uz;
    }
    const float inv = 1.0f / l;
    ox *= inv; oy *= inv; oz *= inv;
    const float* R = affine + (size_t)qi * 12;  // inverse rotation = R^T . o
    o[0] = (__bf16)(R[0] * ox + R[3] * oy + R[6] * oz);
    o[1] = (__bf16)(R[1] * ox + R[4] * oy + R[7] * oz);
    o[2] = (__bf16)(R[2] * ox + R[5] * oy + R[8] * oz);
}

// Rows of at most 16 tokens (the structure encoder's kNN neighbourhoods, vqvae.py:193-263): prep and attention in
// ONE kernel, no Q/K records in HBM.  A block is one row x 16 heads: thread (slot, hh) rotates its token's five
// projected vectors of head hh into the global frame (consecutive hh read consecutive 12-byte vectors of the
// projection), parks the key / value part in LDS and then runs the <= 16-key softmax of its query from there.
constexpr int SK_STRIDE = 16 * 12 + 4;  // floats per head in LDS (+4: spreads the 16 heads over the banks)

template <bool ZF>
__global__ __launch_bounds__(256) void geom_attn_small_kernel(const float* __restrict__ proj,
                                                              const float* __restrict__ affine,
                                                              const uint8_t* __restrict__ amask,
                                                              const int* __restrict__ seq_id,
                                                              const int* __restrict__ chain_id,
                                                              const int* __restrict__ row_cu,
                                                              const float* __restrict__ dist_scale,
                                                              const float* __restrict__ rot_scale,
                                                              __bf16* __restrict__ out, int ld_out, int H) {
    __shared__ __attribute__((aligned(16))) float ks[16 * SK_STRIDE];
    const int slot = threadIdx.x >> 4, hh = threadIdx.x & 15;
    const int h = blockIdx.y * 16 + hh;
    const int r0 = row_cu[blockIdx.x], r1 = row_cu[blockIdx.x + 1];
    const int t = r0 + slot;
    const bool active = t < r1 && h < H;
    float R[9];
    f32x4 qa = f32x4{0.f, 0.f, 0.f, 0.f}, qb = qa;
    int my_seq = 0, my_chain = 0;
    float* kd_ = ks + hh * SK_STRIDE + slot * 12;
    if (active) {
        const float* A = affine + (size_t)t * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = A[k];
        const V3 T{A[9], A[10], A[11]};
        const float* p = proj + (size_t)t * 15 * H;
        auto vec = [&](int v) { return V3{p[v * 3], p[v * 3 + 1], p[v * 3 + 2]}; };
        const float cr = softplus(rot_scale[h]) * (LOG2E * 0.57735026918962576f);
        const float cd = softplus(dist_scale[h]) * (LOG2E * 0.57735026918962576f);
        const V3 qr = rot_apply(R, vec(h)), kr = rot_apply(R, vec(H + h)), vv = rot_apply(R, vec(2 * H + h));
        V3 qd = rot_apply(R, vec(3 * H + h)), kd = rot_apply(R, vec(4 * H + h));
        qd = {qd.x + T.x, qd.y + T.y, qd.z + T.z};
        kd = {kd.x + T.x, kd.y + T.y, kd.z + T.z};
        qa = f32x4{qr.x * cr, qr.y * cr, qr.z * cr, qd.x * cd};
        qb = f32x4{qd.y * cd, qd.z * cd, 0.f, 0.f};
        my_seq = seq_id[t];
        my_chain = chain_id[t];
        *(f32x4*)kd_ = f32x4{kr.x, kr.y, kr.z, kd.x * cd};
        *(f32x4*)(kd_ + 4) = f32x4{kd.y * cd, kd.z * cd, vv.x, vv.y};
        *(f32x4*)(kd_ + 8) = f32x4{vv.z, amask[t] ? 1.f : 0.f, __int_as_float(my_seq), __int_as_float(my_chain)};
    }
    __syncthreads();
    if (!active) return;
    const int nk = r1 - r0;
    float m = -FLT_MAX, l = 0.f, ox = 0.f, oy = 0.f, oz = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
    for (int k = 0; k < nk; ++k) {
        const float* kp = ks + hh * SK_STRIDE + k * 12;
        const f32x4 ka = *(const f32x4*)kp, kb = *(const f32x4*)(kp + 4), kc = *(const f32x4*)(kp + 8);
        const bool ok = (kc[1] != 0.f) && (__float_as_int(kc[3]) == my_chain);
        const float dx = qa[3] - ka[3], dy = qb[0] - kb[0], dz = qb[1] - kb[1];
        float s = qa[0] * ka[0] + qa[1] * ka[1] + qa[2] * ka[2] - sqrtf(dx * dx + dy * dy + dz * dz);
        s += (__float_as_int(kc[2]) == my_seq) ? LOG2E : 0.f;
        if constexpr (!ZF) { ux += kb[2]; uy += kb[3]; uz += kc[0]; }
        if (ok) {
            if (s > m) {
                const float a = __builtin_amdgcn_exp2f(m - s);
                l *= a; ox *= a; oy *= a; oz *= a;
                m = s;
            }
            const float pw = __builtin_amdgcn_exp2f(s - m);
            l += pw;
            ox += pw * kb[2];
            oy += pw * kb[3];
            oz += pw * kc[0];
        }
    }
    __bf16* o = out + (size_t)t * ld_out + h * 3;
    if constexpr (ZF) {
        if (amask[t] == 0 || l == 0.f) {
            o[0] = (__bf16)0.f; o[1] = (__bf16)0.f; o[2] = (__bf16)0.f;
            return;
        }
    } else if (l == 0.f) {
        l = (float)nk; ox = ux; oy = uy; oz = uz;
    }
    const float inv = 1.0f / l;
    ox *= inv; oy *= inv; oz *= inv;
    o[0] = (__bf16)(R[0] * ox + R[3] * oy + R[6] * oz);  // inverse rotation = R^T . o
    o[1] = (__bf16)(R[1] * ox + R[4] * oy + R[7] * oz);
    o[2] = (__bf16)(R[2] * ox + R[5] * oy + R[8] * oz);
}

}  // namespace

extern "C" int esmk_frames(const float* coords, float* affine, uint8_t* mask, int B, int L, esmk_stream_t stream) {
    if (B == 0 || L == 0) return ESMK_OK;
    if (!coords || !affine || !mask) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_frames: null pointer");
    if (B < 0 || L < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_frames: bad sizes");
    if (B == 0 || L == 0) return ESMK_OK;
    hipLaunchKernelGGL(frames_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, coords, affine, mask, L);
    ESMK_CHECK_LAUNCH("esmk_frames");
    return ESMK_OK;
}

extern "C" size_t esmk_geom_scratch_bytes(int M, int v_heads) {
    if (M <= 0 || v_heads <= 0) return 0;
    return (size_t)M * v_heads * 20 * sizeof(float);
}

extern "C" int esmk_geom_attn(const float* proj, const float* affine, const uint8_t* affine_mask, const int32_t* seq_id,
                              const int32_t* chain_id, const int32_t* row_cu, int n_rows, int max_row_len,
                              const float* dist_scale, const float* rot_scale, void* out, int ld_out, void* scratch,
                              int M, int H, esmk_stream_t stream) {
    return esmk_geom_attn_ex(proj, affine, affine_mask, seq_id, chain_id, row_cu, n_rows, max_row_len, dist_scale,
                             rot_scale, out, ld_out, scratch, M, H, 1, stream);
}

extern "C" int esmk_geom_attn_ex(const float* proj, const float* affine, const uint8_t* affine_mask,
                                 const int32_t* seq_id, const int32_t* chain_id, const int32_t* row_cu, int n_rows,
                                 int max_row_len, const float* dist_scale, const float* rot_scale, void* out, int ld_out,
                                 void* scratch, int M, int H, int mask_and_zero_frameless, esmk_stream_t stream) {
    if (M == 0 || n_rows == 0 || max_row_len == 0) return ESMK_OK;
    if (!proj || !affine || !affine_mask || !seq_id || !chain_id || !row_cu || !dist_scale || !rot_scale || !out ||
        (!scratch && max_row_len > 16))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_geom_attn: null pointer");
    if (M < 0 || H <= 0 || n_rows < 0 || max_row_len < 0) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_geom_attn: bad sizes");
    const bool small = max_row_len <= 16;
    if (ld_out < 3 * H || H > 65535 || (!small && n_rows > 65535))
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_geom_attn: ld_out=%d < 3*v_heads=%d or grid too large", ld_out, 3 * H);
    if (M == 0 || n_rows == 0 || max_row_len == 0) return ESMK_OK;
    hipStream_t s = (hipStream_t)stream;
    float* qrec = (float*)scratch;
    float* krec = qrec + (size_t)M * H * 8;
    if (ld_out > 3 * H) {
        if (hipMemsetAsync(out, 0, (size_t)M * ld_out * 2, s) != hipSuccess)
            ESMK_FAIL(ESMK_ERR_LAUNCH, "esmk_geom_attn: memset failed");
    }
    const bool zf = mask_and_zero_frameless != 0;
    if (small) {  // fused prep + attention, no scratch
        const dim3 grid(n_rows, (H + 15) / 16);
        if (zf)
            hipLaunchKernelGGL(geom_attn_small_kernel<true>, grid, dim3(256), 0, s, proj, affine, affine_mask, seq_id, chain_id,
                               row_cu, dist_scale, rot_scale, (__bf16*)out, ld_out, H);
        else
            hipLaunchKernelGGL(geom_attn_small_kernel<false>, grid, dim3(256), 0, s, proj, affine, affine_mask, seq_id, chain_id,
                               row_cu, dist_scale, rot_scale, (__bf16*)out, ld_out, H);
        ESMK_CHECK_LAUNCH("esmk_geom_attn(small rows)");
        return ESMK_OK;
    }
    const size_t total = (size_t)M * H;
    hipLaunchKernelGGL(geom_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, proj, affine,
                       affine_mask, seq_id, chain_id, dist_scale, rot_scale, qrec, krec, M, H);
    ESMK_CHECK_LAUNCH("esmk_geom_attn(prep)");
    {
        const dim3 grid((max_row_len + GQ - 1) / GQ, n_rows, H);
        if (zf)
            hipLaunchKernelGGL(geom_attn_kernel<true>, grid, dim3(GQ), 0, s, qrec, krec, affine, affine_mask, seq_id,
                               chain_id, row_cu, (__bf16*)out, ld_out, M, H);
        else
            hipLaunchKernelGGL(geom_attn_kernel<false>, grid, dim3(GQ), 0, s, qrec, krec, affine, affine_mask, seq_id,
                               chain_id, row_cu, (__bf16*)out, ld_out, M, H);
    }
    ESMK_CHECK_LAUNCH("esmk_geom_attn");
    return ESMK_OK;
}
