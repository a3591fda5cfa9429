// Shared device/host helpers for the gfx950 trunk kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/esmk.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// error plumbing (host)
void esmk_set_error(const char* fmt, ...);
#define ESMK_FAIL(code, ...)         \
    do {                             \
        esmk_set_error(__VA_ARGS__); \
        return (code);               \
    } while (0)
#define ESMK_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) ESMK_FAIL(ESMK_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Experiment switches.  The product library (libesmk.so) compiles every switch to its default: one dispatch path,
// no environment lookups.  Only the experiments build (`make EXPERIMENTS=1` -> libesmk_x.so, used by tools/ and by the
// tests that cover the alternative kernels) reads them from the environment, per call.
#ifdef ESMK_EXPERIMENTS
#include <stdlib.h>
static inline int esmk_switch(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static inline constexpr int esmk_switch(const char*, int dflt) { return dflt; }
#endif

// Kernels that need more than 64 KiB of dynamic LDS must be told so once PER DEVICE (the attribute belongs to the
// device's copy of the code object).  `done` is a per-kernel bit mask of devices already set up; a lost race between
// threads only repeats the (idempotent) call.  Returns false when the runtime refuses.
#include <atomic>
static inline bool esmk_reserve_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return true;
}

// v * a with four scalar multiplies: a vector multiply becomes two v_pk_mul_f32, which issue far slower on gfx950
// than the scalar instructions they replace (MI355X_MICROARCH.md, packed f32 VALU)
__device__ __forceinline__ f32x4 scale4(f32x4 v, float a) {
    f32x4 r;
    r[0] = v[0] * a; r[1] = v[1] * a; r[2] = v[2] * a; r[3] = v[3] * a;
    return r;
}

// ---- bf16 <-> f32 (round-to-nearest-even through the hardware cvt; NaN stays NaN) ----
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) {
    return __uint_as_float(((unsigned int)u) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// one v_cvt_pk_bf16_f32 (a vector conversion: does not depend on the SLP vectoriser pairing two scalar converts)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// XCD-aware bijective remap of a 1-D block id: blocks that are dispatched round-robin over the 8 XCDs
// get contiguous chunks of the logical id space, so neighbours in logical order share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
