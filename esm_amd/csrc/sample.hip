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
