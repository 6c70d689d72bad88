// Probe: can ONE wave overlap independent VALU work with its own in-flight MFMAs?  Loop body = 1 x v_mfma_f32_32x32x16_f16
// (8 passes = 32 cycles) + N independent v_fma_f32 on other registers.  No overlap: 32 + ~4.2 N cycles; overlap: max(32, 4 + 4.2 N).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_valu_overlap_probe.hip -o mfma_valu_overlap_probe && ./mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int N, int NACC>
__global__ void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
    float v[8] = {1.0f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f};
    const float c0 = 0.999f, c1 = 0.001f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j % 8]) : "v"(c0), "v"(c1));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][15];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (float)(iters * NACC);
}
template <int N> void run(float* d) {
    float o[65];
    k<N, 4><<<1, 64>>>(d, 2000); hipMemcpy(o, d, 260, hipMemcpyDeviceToHost);
    printf("1 MFMA (4 accumulators rotating) + %2d independent VALU: %.1f cycles per MFMA\n", N, o[64]);
}
int main() {
    float* d; hipMalloc(&d, 1024);
    run<0>(d); run<2>(d); run<4>(d); run<6>(d); run<8>(d); run<12>(d); run<16>(d); run<24>(d);
    return 0;
}
