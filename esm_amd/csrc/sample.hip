// Fused per-position sampling for the iterative decoding loop around the trunk:
// top-p filter -> invalid-id mask -> temperature softmax + multinomial draw (or argmax) -> log-softmax
// metadata (entropy, log-prob of the chosen token, top probability), one wave per row of V logits held in LDS.
// Mirrors, in this order, what the reference does with several full-size temporaries per track and step:
//   top_p_logits           esm/utils/sampling.py:293-313   (filtered entries := finfo(float32).min, in place)
//   sample_logits          esm/utils/sampling.py:153-200   (invalid ids := -inf in place; T == 0 -> argmax;
//                                                           else multinomial(softmax(logits / T)))
//   _sample_track          esm/utils/generation.py:703-733 (log_softmax of the MODIFIED logits, where(mask, ..))
//   _compute_track_metadata esm/utils/generation.py:786-822 (Categorical entropy, sampled log-prob, top prob)
// torch.multinomial(probs, 1) draws q ~ Exp(1) per entry and returns argmax(probs / q); the caller passes q, so
// a test can feed the very noise torch would have used and compare ids exactly.
#include <float.h>

#include "common.hpp"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ unsigned sort_key(float f) {  // monotone float -> unsigned
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// argmax with first-index tie break across the wave
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

__global__ __launch_bounds__(64) void sample_track_kernel(float* __restrict__ logits, int V,
                                                          const uint8_t* __restrict__ valid, float temperature,
                                                          float top_p, const float* __restrict__ noise,
                                                          const int64_t* __restrict__ tokens,
                                                          const uint8_t* __restrict__ smask,
                                                          int64_t* __restrict__ sampled, float* __restrict__ entropy,
                                                          float* __restrict__ logprob, float* __restrict__ top_prob) {
    extern __shared__ float l[];  // the row
    const int lane = threadIdx.x;
    const size_t r = blockIdx.x;
    float* row = logits + r * V;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) {
        const float x = row[v];
        l[v] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    // ---- top-p: keep the sorted prefix whose cumulative softmax mass is <= top_p, and always the maximum ----
    if (top_p < 1.0f) {
        float z = 0.f;
        for (int v = lane; v < V; v += 64) z += __expf(l[v] - mx);
        z = wave_sum(z);
        unsigned ans = 0;  // largest key K with mass{key >= K} > top_p
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = ans | (1u << bit);
            float s = 0.f;
            for (int v = lane; v < V; v += 64)
                if (sort_key(l[v]) >= cand) s += __expf(l[v] - mx);
            s = wave_sum(s);
            if (s / z > top_p) ans = cand;
        }
        const unsigned kmax = sort_key(mx);
        for (int v = lane; v < V; v += 64) {
            const unsigned k = sort_key(l[v]);
            if (!(k > ans || k == kmax)) l[v] = -FLT_MAX;
        }
    }
    // ---- invalid ids ----
    if (valid)
        for (int v = lane; v < V; v += 64)
            if (!valid[v]) l[v] = -INFINITY;
    mx = -INFINITY;
    for (int v = lane; v < V; v += 64) {
        row[v] = l[v];  // the caller sees the modified logits, as the reference's in-place edits do
        mx = fmaxf(mx, l[v]);
    }
    mx = wave_max(mx);
    // ---- draw ----
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (noise && temperature > 0.f) {
        const float inv_t = 1.0f / temperature;
        float zt = 0.f;
        for (int v = lane; v < V; v += 64) zt += expf((l[v] - mx) * inv_t);  // max(l/T) = max(l)/T for T > 0
        zt = wave_sum(zt);
        const float* q = noise + r * V;
        for (int v = lane; v < V; v += 64) {
            const float sc = (expf((l[v] - mx) * inv_t) / zt) / q[v];
            if (sc > best) { best = sc; bi = v; }
        }
    } else {
        for (int v = lane; v < V; v += 64)
            if (l[v] > best) { best = l[v]; bi = v; }
    }
    wave_argmax(best, bi);
    // ---- metadata on log_softmax(modified logits) ----
    float z = 0.f;
    for (int v = lane; v < V; v += 64) z += expf(l[v] - mx);
    z = wave_sum(z);
    // log_softmax as (l - max) - log(sum): max + log(sum) would swallow the log when every survivor is finfo.min
    const float logz = logf(z);
    float ent = 0.f;
    for (int v = lane; v < V; v += 64) {
        const float lp = (l[v] - mx) - logz;
        const float p = expf(lp);
        ent -= p * fmaxf(lp, -FLT_MAX);  // torch.distributions.Categorical clamps logits at finfo.min
    }
    ent = wave_sum(ent);
    if (lane == 0) {
        const bool on = smask ? smask[r] != 0 : true;
        const int64_t id = on ? (int64_t)bi : tokens[r];
        sampled[r] = id;
        entropy[r] = ent;
        top_prob[r] = expf(-logz);
        logprob[r] = (on && id >= 0 && id < V) ? ((l[id] - mx) - logz) : 0.f;
    }
}

// Which positions of a [B, L] token matrix the sampler may overwrite (one thread per position).
struct SpecialIds { int n; long long id[64]; };
__global__ void sampling_mask_kernel(const int64_t* __restrict__ tokens, int B, int L, SpecialIds sp, long long mask_id,
                                     int only_masked, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int col = i % L;
    const long long t = tokens[i];
    bool on = col != 0 && col != L - 1;          // never the BOS / EOS columns
    for (int k = 0; k < sp.n; ++k) on = on && t != sp.id[k];  // nor a special token (the mask id is not in the list)
    if (only_masked) on = on && t == mask_id;
    out[i] = on ? 1 : 0;
}

// SASA is decoded to a number, not a class: expectation of the bin mid-points under softmax(logits) over the value bins
// (classes 3 .. V-2; class V-1 is the open last bin).  One thread per row; V is small (19).
__global__ void sample_sasa_kernel(float* __restrict__ logits, int R, int V, int L, const uint8_t* __restrict__ valid,
                                   const float* __restrict__ bin_mid, const uint8_t* __restrict__ smask,
                                   float* __restrict__ value, float* __restrict__ entropy) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float* l = logits + (size_t)r * V;
    float mx = -INFINITY;
    int arg = 0;
    for (int v = 0; v < V; ++v) {
        if (valid && !valid[v]) l[v] = -INFINITY;  // in place, as the reference edits its logits
        if (l[v] > mx) { mx = l[v]; arg = v; }     // first maximum, like torch.argmax
    }
    float z = 0.f;
    for (int v = 0; v < V; ++v) z += expf(l[v] - mx);
    const float logz = logf(z);
    float mean = 0.f, ent = 0.f;
    for (int v = 0; v < V; ++v) {
        const float lp = (l[v] - mx) - logz, p = expf(lp);
        if (v >= 3 && v < V - 1) mean += p * bin_mid[v - 3];
        const float pl = p * lp;                   // 0 * -inf is NaN: torch.nan_to_num turns it into 0
        ent -= (pl != pl) ? 0.f : pl;
    }
    const int col = r % L;
    float out = mean;
    if (arg == V - 1 || (smask && !smask[r])) out = INFINITY;  // open last bin / position not sampled: no number
    if (col == 0 || col == L - 1) out = 0.f;                   // BOS / EOS columns
    value[r] = out;
    entropy[r] = ent;
}

}  // namespace

extern "C" int esmk_sample_track(float* logits, int R, int V, const uint8_t* valid, float temperature, float top_p,
                                 const float* noise, const int64_t* tokens, const uint8_t* sampling_mask,
                                 int64_t* sampled, float* entropy, float* logprob, float* top_prob,
                                 esmk_stream_t stream) {
    if (R == 0) return ESMK_OK;
    if (!logits || !sampled || !entropy || !logprob || !top_prob) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_track: null pointer");
    if (sampling_mask && !tokens) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_track: a sampling mask needs the current tokens");
    if (!sampling_mask && tokens) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_track: tokens without a sampling mask");
    if (R < 0 || V <= 0 || V > 16384) ESMK_FAIL(ESMK_ERR_UNSUPPORTED, "esmk_sample_track: need 0 < V <= 16384 (V=%d)", V);
    if (!(temperature >= 0.f) || !(top_p > 0.f)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_track: temperature >= 0 and top_p > 0 required");
    if (temperature > 0.f && !noise) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_track: temperature > 0 needs the Exp(1) noise tensor");
    hipLaunchKernelGGL(sample_track_kernel, dim3(R), dim3(64), (size_t)V * sizeof(float), (hipStream_t)stream, logits, V,
                       valid, temperature, top_p, temperature > 0.f ? noise : nullptr, tokens, sampling_mask, sampled,
                       entropy, logprob, top_prob);
    ESMK_CHECK_LAUNCH("esmk_sample_track");
    return ESMK_OK;
}

extern "C" int esmk_sampling_mask(const int64_t* tokens, int B, int L, const int64_t* special_ids, int n_special,
                                  int64_t mask_id, int only_sample_masked, uint8_t* out, esmk_stream_t stream) {
    if (B == 0 || L == 0) return ESMK_OK;
    if (!tokens || !out || (n_special > 0 && !special_ids)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sampling_mask: null pointer");
    if (B < 0 || L < 0 || n_special < 0 || n_special > 64 || (long long)B * L > 0x7fffffffLL)
        ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sampling_mask: bad sizes (at most 64 special ids)");
    SpecialIds sp{};
    for (int i = 0; i < n_special; ++i)
        if (special_ids[i] != mask_id) sp.id[sp.n++] = special_ids[i];  // HOST array: copied into the launch
    hipLaunchKernelGGL(sampling_mask_kernel, dim3((B * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, tokens, B, L, sp,
                       (long long)mask_id, only_sample_masked, out);
    ESMK_CHECK_LAUNCH("esmk_sampling_mask");
    return ESMK_OK;
}

extern "C" int esmk_sample_sasa(float* logits, int R, int V, int L, const uint8_t* valid, const float* bin_mid,
                                const uint8_t* sampling_mask, float* value, float* entropy, esmk_stream_t stream) {
    if (R == 0) return ESMK_OK;
    if (!logits || !bin_mid || !value || !entropy) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_sasa: null pointer");
    if (R < 0 || V < 5 || V > 4096 || L <= 0 || (R % L)) ESMK_FAIL(ESMK_ERR_INVALID_ARG, "esmk_sample_sasa: bad sizes (R = B * L rows, V >= 5)");
    hipLaunchKernelGGL(sample_sasa_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, R, V, L, valid,
                       bin_mid, sampling_mask, value, entropy);
    ESMK_CHECK_LAUNCH("esmk_sample_sasa");
    return ESMK_OK;
}
