// Probe: v_mfma_f32_16x16x16_f16 operand layout, fed from the 32-column "row layout" through v_permlane16_swap on
// packed-half VGPRs, with the 3-term split (hi*hi + hi*lo + lo*hi); plus its issue rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma16h_probe.hip -o mfma16h_probe && ./mfma16h_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ void swap16(float& x, float& y) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y)); }

// A: [16 rows][16 k] fp32, B: [16 k][32 cols] fp32, D: [16][32]
__global__ void k_probe(const float* A, const float* B, float* D) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    // row layout: this lane holds k = 8h + jj of column col
    h8 bh, bl;
    for (int jj = 0; jj < 8; ++jj) {
        const float x = B[(8 * h + jj) * 32 + col];
        const _Float16 hh = (_Float16)x;
        bh[jj] = hh; bl[jj] = (_Float16)(x - (float)hh);
    }
    f32x4 fh = __builtin_bit_cast(f32x4, bh), fl = __builtin_bit_cast(f32x4, bl);
    float p[4] = {fh[0], fh[1], fh[2], fh[3]}, q[4] = {fl[0], fl[1], fl[2], fl[3]};
    swap16(p[0], p[2]); swap16(p[1], p[3]);
    swap16(q[0], q[2]); swap16(q[1], q[3]);
    bh = __builtin_bit_cast(h8, f32x4{p[0], p[1], p[2], p[3]}); bl = __builtin_bit_cast(h8, f32x4{q[0], q[1], q[2], q[3]});
    // A operand: lane (o = lane & 15, kq = lane >> 4) holds k = 4 kq + i
    h4 ah, al;
    for (int i = 0; i < 4; ++i) {
        const float w = A[(lane & 15) * 16 + 4 * (lane >> 4) + i];
        const _Float16 hh = (_Float16)w;
        ah[i] = hh; al[i] = (_Float16)(w - (float)hh);
    }
    for (int S = 0; S < 2; ++S) {
        h4 xh, xl;
        for (int i = 0; i < 4; ++i) { xh[i] = bh[4 * S + i]; xl[i] = bl[4 * S + i]; }
        f32x4 acc = {0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al, xh, acc, 0, 0, 0);
        for (int i = 0; i < 4; ++i) D[(4 * (lane >> 4) + i) * 32 + 16 * S + (lane & 15)] = acc[i];
    }
}

template <int NACC>
__global__ void k_rate(float* out, int iters) {
    h4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc[n], 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (float)(iters * NACC);
}

int main(int argc, char** argv) {
    std::vector<float> A(256), B(512), D(512);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    const float sa = argc > 1 ? atof(argv[1]) : 1.0f, sb = argc > 2 ? atof(argv[2]) : 3.0f;
    for (auto& v : A) v = rnd() * sa;
    for (auto& v : B) v = rnd() * sb;
    float *dA, *dB, *dD, *dO;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 2048); hipMalloc(&dD, 2048); hipMalloc(&dO, 1024);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0, ref_max = 0;
    for (int o = 0; o < 16; ++o) for (int c = 0; c < 32; ++c) {
        double r = 0;
        for (int k = 0; k < 16; ++k) r += (double)A[o * 16 + k] * B[k * 32 + c];
        err = fmax(err, fabs(r - D[o * 32 + c])); ref_max = fmax(ref_max, fabs(r));
    }
    printf("layout+split: max abs err %.3e (max |ref| %.3e) rel %.3e\n", err, ref_max, err / ref_max);
    float o[65];
    k_rate<1><<<1, 64>>>(dO, 4096); hipMemcpy(o, dO, 260, hipMemcpyDeviceToHost); printf("16x16x16 f16, 1 acc (dependent): %.1f cycles(memtime units)\n", o[64]);
    k_rate<4><<<1, 64>>>(dO, 4096); hipMemcpy(o, dO, 260, hipMemcpyDeviceToHost); printf("16x16x16 f16, 4 acc rotated: %.1f\n", o[64]);
    k_rate<8><<<1, 64>>>(dO, 4096); hipMemcpy(o, dO, 260, hipMemcpyDeviceToHost); printf("16x16x16 f16, 8 acc rotated: %.1f\n", o[64]);
    return err < 1e-4 * ref_max ? 0 : 1;
}
