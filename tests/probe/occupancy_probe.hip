// Probe: VALU / MFMA throughput of ONE SIMD with one wave vs two waves resident (is a single wave able to saturate the VALU?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 occupancy_probe.hip -o occupancy_probe && ./occupancy_probe
// A block of 64*W threads puts W/4 waves on each SIMD of one CU (W = 4: one per SIMD, W = 8: two per SIMD).  Every wave runs the
// same instruction stream; reported: cycles per instruction PER WAVE (so two waves at the same figure = twice the throughput).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define BODY(NAME, ASM, NINS)                                                                              \
    __global__ void NAME(float* out, int iters) {                                                         \
        float a = threadIdx.x * 0.01f, b = 1.5f, c = 0.25f, d = 2.0f, e = 3.0f, f = 0.5f, g = 0.1f, h = 0.7f; \
        __syncthreads();                                                                                   \
        const long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < iters; ++it) { asm volatile(REP64(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)); } \
        const long long t1 = __builtin_readcyclecounter();                                                \
        out[threadIdx.x] = a + b + c + d + e + f + g + h;                                                 \
        if ((threadIdx.x & 63) == 0) out[1024 + (threadIdx.x >> 6)] = (float)(t1 - t0) / (float)(iters * 64 * NINS); \
    }
BODY(k_fma, "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t", 4)
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\tv_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %4, %5\n\t", 4)
BODY(k_exp, "v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\t", 4)
BODY(k_salu, "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1\n\t", 4)
BODY(k_mix, "v_fma_f32 %0, %4, %5, %0\n\ts_add_u32 s20, s20, 1\n\tv_fma_f32 %1, %4, %5, %1\n\ts_nop 0\n\t", 4)

// MFMA stream with F independent VALU fillers per MFMA (4 accumulators rotate)
template <int F>
__global__ void k_mfma(float* out, int iters) {
    f32x16 acc[4] = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < F; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]), "v"(v[(k + 2) & 7]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[1024 + (threadIdx.x >> 6)] = (float)(t1 - t0) / (float)(iters * 16);
}

int main() {
    float* d; hipMalloc(&d, 8192); float o[2048];
    const int Ws[] = {1, 4, 8, 12, 16};
#define RUN(K, LABEL) for (int W : Ws) { K<<<1, 64 * W>>>(d, 200); hipMemcpy(o, d, 8192, hipMemcpyDeviceToHost); \
        float mx = 0; for (int w = 0; w < W; ++w) mx = o[1024 + w] > mx ? o[1024 + w] : mx; printf("%-34s waves/CU %2d : %.2f cycles per unit per wave\n", LABEL, W, mx); }
    RUN(k_fma, "v_fma_f32") RUN(k_cvtpk, "v_cvt_pk_f16_f32") RUN(k_exp, "v_exp_f32") RUN(k_salu, "s_add_u32") RUN(k_mix, "fma/s_add/fma/s_nop")
    RUN(k_mfma<0>, "mfma 32x32x16 f16 + 0 fma (per MFMA)") RUN(k_mfma<4>, "mfma + 4 fma (per MFMA)") RUN(k_mfma<8>, "mfma + 8 fma (per MFMA)")
    RUN(k_mfma<12>, "mfma + 12 fma (per MFMA)") RUN(k_mfma<16>, "mfma + 16 fma (per MFMA)")
    return 0;
}
