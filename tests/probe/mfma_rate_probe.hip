// GPU probe: issue rate of v_mfma_f32_32x32x2_f32 with 1/2/4 independent accumulators, with and without a
// VALU-produced B operand, one wave per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o p && ./p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int VALU> __global__ __launch_bounds__(64, 1) void k(float* out, unsigned long long* cyc, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float bb = b;
            if (VALU) { bb = b * a + (float)u; b = bb * 0.999f; }
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[u % NACC], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int VALU> void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(64), 0, 0, out, cyc, 1.0f, 0.5f);
        hipDeviceSynchronize();
        unsigned long long h[1024];
        hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < grid; ++i) m += h[i];
        printf("%-28s grid %4d: %.1f cycles/MFMA\n", name, grid, m / grid / (256.0 * 16));
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 0>("1 acc (dependent chain)");
    run<2, 0>("2 acc");
    run<4, 0>("4 acc");
    run<1, 1>("1 acc + VALU B");
    run<4, 1>("4 acc + VALU B");
    return 0;
}
