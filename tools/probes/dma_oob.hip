// Probe: what does an LDS-DMA buffer load write for lanes that the descriptor's range check rejects?
// LDS is pre-filled with 0x7fc07fc0 (bf16 NaN pairs); rows >= rows_in are out of range.
//   hipcc --offload-arch=gfx950 -O2 -o dma_oob dma_oob.hip && ./dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void probe(const unsigned* src, unsigned* dst, int nrec, int use_soffset) {
    __shared__ __attribute__((aligned(16))) unsigned smem[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) smem[lane * 4 + i] = 0x7fc07fc0u;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    // lane l loads 16 bytes at byte offset l*64 (+ 1024 through soffset when asked)
    if (use_soffset) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem), 16, lane * 64, 1024, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem), 16, lane * 64, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) dst[lane * 4 + i] = smem[lane * 4 + i];
}
int main() {
    const int n = 4096;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x10000u + i;
    unsigned *src, *dst;
    hipMalloc(&src, n * 4); hipMalloc(&dst, 256 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> out(256);
    for (int mode = 0; mode < 2; ++mode) {
        // nrec = 10 * 64 + 16: lanes 0..10 in range without soffset
        const int nrec = mode == 0 ? 10 * 64 + 16 : 1024 + 10 * 64 + 16;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, dst, nrec, mode);
        hipMemcpy(out.data(), dst, 256 * 4, hipMemcpyDeviceToHost);
        printf("mode %d (soffset %s) nrec %d\n", mode, mode ? "1024" : "0", nrec);
        for (int l : {0, 9, 10, 11, 12, 26, 27, 63}) printf("  lane %2d: %08x %08x %08x %08x\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
    return 0;
}
