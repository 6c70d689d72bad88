// Probe: v_mfma_f32_16x16x32_f16 operand layout as dedf_edge16.h assumes it:
//   A[row = lane & 15][k = 8 (lane >> 4) + j],  B[k = 8 (lane >> 4) + j][col = lane & 15],  D[row = 4 (lane >> 4) + r][col = lane & 15]
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma16x32_probe.hip -o mfma16x32_probe && ./mfma16x32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void k_probe(const float* A, const float* B, float* D) {      // A [16][32], B [32][16], D [16][16]
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[c * 32 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 16 + c]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}
template <int NACC> __global__ void k_rate(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][3];
    out[threadIdx.x + blockDim.x * blockIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[4096] = (float)(t1 - t0) / (float)(iters * NACC);
}
int main() {
    std::vector<float> A(512), B(512), D(256);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)((s >> 8) & 0xff) - 128) / 64.0f; };      // exactly representable in fp16
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *dD, *dO;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024); hipMalloc(&dO, 4097 * 4 + 1024 * 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int o = 0; o < 16; ++o) for (int c = 0; c < 16; ++c) {
        double r = 0;
        for (int k = 0; k < 32; ++k) r += (double)A[o * 32 + k] * B[k * 16 + c];
        err = fmax(err, fabs(r - D[o * 16 + c]));
    }
    printf("16x16x32 f16 layout: max abs err %.3e (%s)\n", err, err < 1e-3 ? "layout as assumed" : "LAYOUT DIFFERS");
    float o;
    k_rate<4><<<1, 64>>>(dO, 4096); hipMemcpy(&o, dO + 4096, 4, hipMemcpyDeviceToHost); printf("1 wave, 4 acc rotated: %.1f cycles per MFMA\n", o);
    k_rate<4><<<1, 128>>>(dO, 4096); hipMemcpy(&o, dO + 4096, 4, hipMemcpyDeviceToHost); printf("2 waves on a CU (different SIMDs), 4 acc: %.1f\n", o);
    k_rate<4><<<1, 512>>>(dO, 4096); hipMemcpy(&o, dO + 4096, 4, hipMemcpyDeviceToHost); printf("8 waves on a CU (2 per SIMD), 4 acc: %.1f cycles per MFMA per wave\n", o);
    return err < 1e-3 ? 0 : 1;
}
