// GPU probe: MFMA issue rate when every K-group's A operand (1 KiB per wave per NMF MFMAs) streams from an L2-resident
// weight image through a buffer descriptor, prefetched PD groups ahead — the access pattern of the fused edge kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NMF, int PD> __global__ __launch_bounds__(64, 1) void k(const float* w, unsigned wbytes, float* out, unsigned long long* cyc, float b0, int ngroups) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, wbytes, 0x00020000);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    const int lane16 = (threadIdx.x & 63) * 16;
    f32x4 ring[PD];
    int soff = (blockIdx.x * 7919 % 128) * 1024;      // waves start at different places of the image
    const int wrap = wbytes - 1024;
    for (int p = 0; p < PD; ++p) { ring[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, soff, 0)); soff = soff + 1024 > wrap ? 0 : soff + 1024; }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int g = 0; g < ngroups; g += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const f32x4 a = ring[p];
            ring[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, soff, 0));
            soff = soff + 1024 > wrap ? 0 : soff + 1024;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NMF; ++u) acc[u % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % 4], b0, acc[u % 4], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NMF, int PD> void run() {
    const unsigned wbytes = 560 * 1024;
    float *w, *out; unsigned long long* cyc;
    (void)hipMalloc(&w, wbytes); (void)hipMemset(w, 0, wbytes); (void)hipMalloc(&out, 2048 * 64 * 4); (void)hipMalloc(&cyc, 2048 * 8);
    for (int grid : {1, 256, 1024}) {
        const int ng = 4096;
        hipLaunchKernelGGL((k<NMF, PD>), dim3(grid), dim3(64), 0, 0, w, wbytes, out, cyc, 0.5f, ng);
        (void)hipDeviceSynchronize();
        unsigned long long h[2048];
        (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
        printf("MFMAs per 1KiB load %2d, prefetch %2d groups, grid %4d: %6.1f cycles/MFMA  (%.2f B/cycle/wave)\n", NMF, PD, grid, m / grid / (double)(ng * NMF), 1024.0 * ng / (m / grid));
    }
    (void)hipFree(w); (void)hipFree(out); (void)hipFree(cyc);
}
int main() { run<4, 8>(); run<4, 16>(); run<8, 8>(); run<16, 8>(); return 0; }
