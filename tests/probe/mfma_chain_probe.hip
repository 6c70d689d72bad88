// GPU probe (test infrastructure): validates the v_mfma_f32_32x32x2_f32 fragment maps and the
// register-chained "transposed" MLP scheme of csrc/dedf_layout.h against a host fp64 reference.
//   hipcc --offload-arch=gfx950 -O2 -I diffusion_edf_amd/csrc tests/probe/mfma_chain_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "dedf_layout.h"
using namespace dedf;
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

// x: [32 items][64] ; layer1 64->128 (+bias, silu) ; layer2 128->64 ; out [32][64]
__global__ void chain(const float* __restrict__ x, const float4* __restrict__ A1, const float* __restrict__ b1,
                      const float4* __restrict__ A2, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    f32x16 h[4];
    for (int To = 0; To < 4; ++To) {
        for (int r = 0; r < 16; ++r) h[To][r] = b1[(To * 2 + hi) * 16 + r];
        // 32 steps: step s supplies k = s + 32*hi  (the first-layer convention)
        for (int g = 0; g < 8; ++g) {
            float4 a = A1[(To * 8 + g) * 64 + lane];
            float av[4] = {a.x, a.y, a.z, a.w};
            for (int j = 0; j < 4; ++j) {
                float b = x[col * 64 + (4 * g + j) + 32 * hi];
                h[To] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b, h[To], 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) { float v = h[To][r]; h[To][r] = v / (1.0f + expf(-v)); }
    }
    for (int To = 0; To < 2; ++To) {
        f32x16 acc = {0};
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 a = A2[(To * 16 + T * 4 + g) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, h[T][4 * g + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, h[T][4 * g + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, h[T][4 * g + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, h[T][4 * g + 3], acc, 0, 0, 0);
            }
        for (int r = 0; r < 16; ++r) out[col * 64 + To * 32 + rowmap(r, hi)] = acc[r];
    }
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(32 * 64), W1(128 * 64), b1(128), W2(64 * 128);
    for (auto& v : x) v = nd(rng);
    for (auto& v : W1) v = nd(rng) / 8;
    for (auto& v : b1) v = nd(rng);
    for (auto& v : W2) v = nd(rng) / 11;
    std::vector<KStep> s1;
    for (int s = 0; s < 32; ++s) s1.push_back({s, s + 32});
    auto A1 = pack_A(128, s1, [&](int o, int k) { return W1[o * 64 + k]; });
    auto A2 = pack_A(64, chain_steps(128), [&](int o, int k) { return W2[o * 128 + k]; });
    auto B1 = pack_rows(128, [&](int o) { return b1[o]; });
    float *dx, *dA1, *db1, *dA2, *dout;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dA1, A1.size() * 4)); CK(hipMalloc(&db1, B1.size() * 4));
    CK(hipMalloc(&dA2, A2.size() * 4)); CK(hipMalloc(&dout, 32 * 64 * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA1, A1.data(), A1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, B1.data(), B1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dx, (const float4*)dA1, db1, (const float4*)dA2, dout);
    CK(hipDeviceSynchronize());
    std::vector<float> out(32 * 64);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int c = 0; c < 32; ++c) {
        double h[128];
        for (int o = 0; o < 128; ++o) {
            double a = b1[o];
            for (int k = 0; k < 64; ++k) a += (double)W1[o * 64 + k] * x[c * 64 + k];
            h[o] = a / (1 + std::exp(-a));
        }
        for (int o = 0; o < 64; ++o) {
            double a = 0;
            for (int k = 0; k < 128; ++k) a += (double)W2[o * 128 + k] * h[k];
            maxerr = std::max(maxerr, std::abs(a - out[c * 64 + o]));
            maxref = std::max(maxref, std::abs(a));
        }
    }
    printf("mfma chain probe: max abs err %.3e (max |ref| %.3f) -> %s\n", maxerr, maxref, maxerr < 1e-4 * maxref ? "OK" : "FAIL");
    return maxerr < 1e-4 * maxref ? 0 : 1;
}
