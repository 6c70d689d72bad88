// GPU probe: 3-term split-fp16 MFMA (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16) in the register-chained transposed
// layout: layer 1 (64 -> 128, +bias, SiLU) and layer 2 (128 -> 64), compared with an fp64 host reference and with the error
// an exact-fp32 evaluation makes.  K-chunk = 8 accumulator registers of each half-wave = 16 k's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
static inline int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ inline int d_rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ inline void split8(const float* x, h8& hi, h8& lo) {
    for (int j = 0; j < 8; ++j) { const _Float16 h = (_Float16)x[j]; hi[j] = h; lo[j] = (_Float16)(x[j] - (float)h); }
}
__device__ inline f32x16 mfma3(h8 ah, h8 al, h8 bh, h8 bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return c;
}
// images: [To][chunk][lane][8] halves, hi and lo
__global__ void k(const float* x, const h8* A1h, const h8* A1l, const float* b1, const h8* A2h, const h8* A2l, float* out) {
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    float xb[32];
    for (int s = 0; s < 32; ++s) xb[s] = x[col * 64 + s + 32 * hi];          // first layer: k = s + 32 hi
    f32x16 h[4];
    for (int To = 0; To < 4; ++To) {
        for (int r = 0; r < 16; ++r) h[To][r] = b1[To * 32 + d_rowmap(r, hi)];
        for (int c = 0; c < 4; ++c) {
            h8 bh, bl; split8(&xb[8 * c], bh, bl);
            h[To] = mfma3(A1h[(To * 4 + c) * 64 + lane], A1l[(To * 4 + c) * 64 + lane], bh, bl, h[To]);
        }
        for (int r = 0; r < 16; ++r) { const float v = h[To][r]; h[To][r] = v / (1.0f + expf(-v)); }
    }
    for (int To = 0; To < 2; ++To) {
        f32x16 acc = {0};
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float t[8];
                for (int j = 0; j < 8; ++j) t[j] = h[T][8 * c + j];
                h8 bh, bl; split8(t, bh, bl);
                acc = mfma3(A2h[(To * 8 + T * 2 + c) * 64 + lane], A2l[(To * 8 + T * 2 + c) * 64 + lane], bh, bl, acc);
            }
        for (int r = 0; r < 16; ++r) out[col * 64 + To * 32 + d_rowmap(r, hi)] = acc[r];
    }
}
template <class KOf> static void pack(int O, int nch, const std::vector<float>& W, int ld, KOf kof, std::vector<_Float16>& H, std::vector<_Float16>& Lo) {
    const int nTo = (O + 31) / 32;
    H.assign((size_t)nTo * nch * 64 * 8, (_Float16)0); Lo = H;
    for (int To = 0; To < nTo; ++To) for (int c = 0; c < nch; ++c) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        const int o = To * 32 + (lane & 31), kk = kof(c, j, lane >> 5);
        const float w = W[(size_t)o * ld + kk];
        const _Float16 hh = (_Float16)w;
        H[(((size_t)To * nch + c) * 64 + lane) * 8 + j] = hh;
        Lo[(((size_t)To * nch + c) * 64 + lane) * 8 + j] = (_Float16)(w - (float)hh);
    }
}
int main() {
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x(32 * 64), W1(128 * 64), b1(128), W2(64 * 128);
    for (auto& v : x) v = nd(rng) * 3; for (auto& v : W1) v = nd(rng) / 8; for (auto& v : b1) v = nd(rng); for (auto& v : W2) v = nd(rng) / 11;
    std::vector<_Float16> A1h, A1l, A2h, A2l;
    pack(128, 4, W1, 64, [](int c, int j, int h) { return 8 * c + j + 32 * h; }, A1h, A1l);
    pack(64, 8, W2, 128, [](int c, int j, int h) { return 32 * (c / 2) + rowmap(8 * (c % 2) + j, h); }, A2h, A2l);
    float *dx, *db1, *dout; _Float16 *d1h, *d1l, *d2h, *d2l;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&db1, 512); (void)hipMalloc(&dout, 32 * 64 * 4);
    (void)hipMalloc(&d1h, A1h.size() * 2); (void)hipMalloc(&d1l, A1h.size() * 2); (void)hipMalloc(&d2h, A2h.size() * 2); (void)hipMalloc(&d2l, A2h.size() * 2);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db1, b1.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(d1h, A1h.data(), A1h.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d1l, A1l.data(), A1l.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d2h, A2h.data(), A2h.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d2l, A2l.data(), A2l.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, (const h8*)d1h, (const h8*)d1l, db1, (const h8*)d2h, (const h8*)d2l, dout);
    std::vector<float> out(32 * 64);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double e16 = 0, e32 = 0, mref = 0;
    for (int c = 0; c < 32; ++c) {
        double h[128]; float hf[128];
        for (int o = 0; o < 128; ++o) {
            double a = b1[o]; float af = b1[o];
            for (int kk = 0; kk < 64; ++kk) { a += (double)W1[o * 64 + kk] * x[c * 64 + kk]; af = fmaf(W1[o * 64 + kk], x[c * 64 + kk], af); }
            h[o] = a / (1 + std::exp(-a)); hf[o] = af / (1.0f + expf(-af));
        }
        for (int o = 0; o < 64; ++o) {
            double a = 0; float af = 0;
            for (int kk = 0; kk < 128; ++kk) { a += (double)W2[o * 128 + kk] * h[kk]; af = fmaf(W2[o * 128 + kk], hf[kk], af); }
            e16 = fmax(e16, fabs(a - out[c * 64 + o])); e32 = fmax(e32, fabs(a - af)); mref = fmax(mref, fabs(a));
        }
    }
    printf("split-fp16x3 chain: max abs err %.3e ; exact fp32 chain: %.3e ; max |ref| %.3f -> %s\n", e16, e32, mref, e16 < 2e-5 * mref ? "OK" : "FAIL");
    return e16 < 2e-5 * mref ? 0 : 1;
}
