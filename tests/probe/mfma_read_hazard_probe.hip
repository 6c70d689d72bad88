// Probe: is the number of wait states hipcc (ROCm 7.2) leaves between v_mfma_f32_32x32x16_f16 and a VALU read of its result
// enough on gfx950?  Runs the same MFMA chain, reads the accumulator (a) the way the compiler schedules it and (b) after an
// additional s_nop 15 + s_nop 15, and counts differing lanes/registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_read_hazard_probe.hip -o mfma_read_hazard_probe && ./mfma_read_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <bool PAD, int NCHAIN>
__global__ void k(const float* in, float* out) {
    h8 a, b, b2;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[threadIdx.x * 8 + i]; b[i] = (_Float16)in[512 + threadIdx.x * 8 + i]; b2[i] = (_Float16)in[1024 + threadIdx.x * 8 + i]; }
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
    // same shape as the failing code: acc1, acc2, acc1 back to back, results read right away
#pragma unroll
    for (int n = 0; n < NCHAIN; ++n) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc2, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b2, acc1, 0, 0, 0);
    }
    if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc1), "+v"(acc2));
    float s = 0.0f;
#pragma unroll
    for (int r = 15; r >= 0; --r) s += acc2[r] * (float)(r + 1) + acc1[r];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
    out[threadIdx.x] = s + acc0[0] * 0.0f;
}

int main() {
    float h[1536];
    unsigned s = 7;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    float *din, *d0, *d1; hipMalloc(&din, sizeof(h)); hipMalloc(&d0, 256); hipMalloc(&d1, 256);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    int bad = 0;
    for (int rep = 0; rep < 200; ++rep) {
        k<false, 1><<<1, 64>>>(din, d0); k<true, 1><<<1, 64>>>(din, d1);
        float a[64], b[64]; hipMemcpy(a, d0, 256, hipMemcpyDeviceToHost); hipMemcpy(b, d1, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) bad += a[i] != b[i];
    }
    printf("chain 1: %d differing lane results over 200 runs\n", bad);
    bad = 0;
    for (int rep = 0; rep < 200; ++rep) {
        k<false, 4><<<1, 64>>>(din, d0); k<true, 4><<<1, 64>>>(din, d1);
        float a[64], b[64]; hipMemcpy(a, d0, 256, hipMemcpyDeviceToHost); hipMemcpy(b, d1, 256, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) bad += a[i] != b[i];
    }
    printf("chain 4: %d differing lane results over 200 runs\n", bad);
    return 0;
}
