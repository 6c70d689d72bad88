// Probe: issue cost (cycles per instruction, one wave) of the VALU instructions the split-fp16 path leans on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 valu_rate_probe.hip -o valu_rate_probe && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define BODY(NAME, ASM)                                                                                   \
    __global__ void NAME(float* out, int iters) {                                                         \
        float a = threadIdx.x * 0.01f, b = 1.5f, c = 0.25f, d = 2.0f, e = 3.0f, f = 0.5f, g = 0.1f, h = 0.7f; \
        const long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < iters; ++it) { asm volatile(REP64(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)); } \
        const long long t1 = __builtin_readcyclecounter();                                                \
        out[threadIdx.x] = a + b + c + d + e + f + g + h;                                                 \
        if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (float)(iters * 64);                           \
    }
// independent destinations rotate over 4 registers so that no instruction depends on the previous one
BODY(k_fma, "v_fma_f32 %0, %4, %5, %0\n\tv_fma_f32 %1, %4, %5, %1\n\tv_fma_f32 %2, %4, %5, %2\n\tv_fma_f32 %3, %4, %5, %3\n\t")
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\tv_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %4, %5\n\t")
BODY(k_cvtf32, "v_cvt_f32_f16 %0, %4\n\tv_cvt_f32_f16 %1, %5\n\tv_cvt_f32_f16 %2, %6\n\tv_cvt_f32_f16 %3, %7\n\t")
BODY(k_cvtf32s, "v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t")
BODY(k_mixlo, "v_fma_mixlo_f16 %0, %4, %5, %6\n\tv_fma_mixlo_f16 %1, %4, %5, %6\n\tv_fma_mixlo_f16 %2, %4, %5, %6\n\tv_fma_mixlo_f16 %3, %4, %5, %6\n\t")
BODY(k_swap, "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t")
BODY(k_accrd, "v_accvgpr_write_b32 a0, %4\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_write_b32 a1, %5\n\tv_accvgpr_read_b32 %1, a1\n\t")
BODY(k_mov, "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t")
BODY(k_exp, "v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\t")

__global__ void k_pk(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {threadIdx.x * 0.01f, 1.0f}, b = {1.5f, 0.5f}, c = {0.25f, 0.3f}, d = {2.0f, 1.0f}, e = {1.0001f, 0.9999f}, f = {0.001f, 0.002f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile(REP64("v_pk_fma_f32 %0, %4, %5, %0\n\tv_pk_fma_f32 %1, %4, %5, %1\n\tv_pk_mul_f32 %2, %4, %2\n\tv_pk_mul_f32 %3, %4, %3\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a[0] + b[1] + c[0] + d[1];
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (float)(iters * 64);
}

int main() {
    float* d; hipMalloc(&d, 1024); float o[65];
#define RUN(K, N) K<<<1, 64>>>(d, 200); hipMemcpy(o, d, 260, hipMemcpyDeviceToHost); printf("%-28s %.2f cycles / instruction\n", #K, o[64] / N);
    RUN(k_fma, 4) RUN(k_pk, 4) RUN(k_cvtpk, 4) RUN(k_cvtf32, 4) RUN(k_cvtf32s, 4) RUN(k_mixlo, 4) RUN(k_swap, 4) RUN(k_accrd, 4) RUN(k_mov, 4) RUN(k_exp, 4)
    return 0;
}
