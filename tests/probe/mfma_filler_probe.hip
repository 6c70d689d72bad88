// Probe (round 6): WHICH instructions of k_edge's VALU stream run under an in-flight v_mfma_f32_32x32x16_f16 of the same (lone) wave?
// Loop body = 1 MFMA (4 accumulators rotating, AGPR or VGPR accumulators) + N fillers of one kind on registers the MFMA does not touch.
// Overlap: max(32, 4 + c N) cycles per MFMA;  no overlap: 32 + c N.   One wave per SIMD (64-thread block, 512 registers requested via
// __launch_bounds__(64, 1)), like the fused kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 mfma_filler_probe.hip -o mfma_filler_probe && ./mfma_filler_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { FMA = 0, MUL, CVTPK, MIXLO, MIXHI, ACCRD, ACCWR, DPP, EXP, PKMUL, MOV, LDSRD, CNDMASK, BUFLD, SPLIT3, SNOP, N_KIND };
static const char* kNames[N_KIND] = {"v_fma_f32", "v_mul_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
                                     "v_fmac_f32_dpp row_shr:1", "v_exp_f32", "v_pk_mul_f32", "v_mov_b32", "ds_read_b128", "v_cndmask_b32", "buffer/global_load_dwordx4",
                                     "split triple (cvt_pk + mixlo + mixhi)", "s_nop 0"};

template <int KIND>
__device__ __forceinline__ void filler(float (&v)[8], int j, float c0, float c1, const f32x4* lds, const f32x4* g, f32x4 (&sink)[4], float& spare) {
    float& x = v[j % 8];
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c0), "v"(c1));
    else if constexpr (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c0));
    else if constexpr (KIND == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(c0), "v"(c1));
    else if constexpr (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c0), "v"(c1));
    else if constexpr (KIND == MIXHI) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c0), "v"(c1));
    else if constexpr (KIND == ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(spare));
    else if constexpr (KIND == ACCWR) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(spare) : "v"(c0));
    else if constexpr (KIND == DPP) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(c0));
    else if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(c1));
    else if constexpr (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&v[2 * (j % 4)])) : "v"(*reinterpret_cast<const double*>(&sink[0])));
    else if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c0));
    else if constexpr (KIND == LDSRD) sink[j % 4] = lds[j % 4 * 64];
    else if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(c0), "v"(c1));
    else if constexpr (KIND == BUFLD) sink[j % 4] = __builtin_nontemporal_load(g + (j % 4) * 64);
    else if constexpr (KIND == SPLIT3) {
        float hi, lo;
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(hi), "=&v"(lo) : "v"(c0), "v"(x));
        x = lo; v[(j + 1) % 8] = hi;
    } else if constexpr (KIND == SNOP) asm volatile("s_nop 0");
}

template <int KIND, int N, bool VGPR_ACC, bool SAME_ACC>
__global__ __launch_bounds__(64, 1) void k(float* out, const f32x4* g, int iters) {
    __shared__ f32x4 lds[4 * 64 + 64];
    lds[threadIdx.x] = f32x4{1, 2, 3, 4};
    __syncthreads();
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;
    float v[8] = {1.0f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f};
    float c0 = 0.999f, c1 = 0.001f, spare = 2.0f;
    asm volatile("" : "+v"(c0), "+v"(c1));
    f32x4 sink[4] = {};
    const f32x4* lp = lds + threadIdx.x;
    const f32x4* gp = g + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            constexpr int dummy = 0;
            const int m = SAME_ACC ? 0 : n;
            if constexpr (VGPR_ACC) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[SAME_ACC ? 0 : n]) : "v"(a), "v"(b));
            } else {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[SAME_ACC ? 0 : n]) : "v"(a), "v"(b));
            }
            (void)m; (void)dummy;
#pragma unroll
            for (int j = 0; j < N; ++j) filler<KIND>(v, j, c0, c1, lp, gp, sink, spare);
        }
        if constexpr (KIND == LDSRD || KIND == BUFLD) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(sink[0]), "+v"(sink[1]), "+v"(sink[2]), "+v"(sink[3])); }
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = spare;
    for (int n = 0; n < 4; ++n) s += acc[n][0] + acc[n][15];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += sink[j][0] + sink[j][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (float)(iters * 4);
}

template <int KIND, int N, bool VA, bool SA> float run1(float* d, const f32x4* g) {
    float o[65];
    k<KIND, N, VA, SA><<<1, 64>>>(d, g, 2000);
    hipMemcpy(o, d, 260, hipMemcpyDeviceToHost);
    return o[64];
}
template <int KIND, bool VA = false, bool SA = false> void run(float* d, const f32x4* g) {
    printf("%-42s %s%s  N=0/2/4/5/6/8/12/16: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f cycles per MFMA\n", kNames[KIND], VA ? "[VGPR acc]" : "[AGPR acc]", SA ? "[same acc]" : "",
           run1<KIND, 0, VA, SA>(d, g), run1<KIND, 2, VA, SA>(d, g), run1<KIND, 4, VA, SA>(d, g), run1<KIND, 5, VA, SA>(d, g), run1<KIND, 6, VA, SA>(d, g), run1<KIND, 8, VA, SA>(d, g),
           run1<KIND, 12, VA, SA>(d, g), run1<KIND, 16, VA, SA>(d, g));
}
int main() {
    float* d; hipMalloc(&d, 1024);
    f32x4* g; hipMalloc(&g, 64 * 1024); hipMemset(g, 0, 64 * 1024);
    run<FMA>(d, g); run<MUL>(d, g); run<CVTPK>(d, g); run<MIXLO>(d, g); run<MIXHI>(d, g); run<SPLIT3>(d, g); run<ACCRD>(d, g); run<ACCWR>(d, g); run<DPP>(d, g); run<EXP>(d, g);
    run<PKMUL>(d, g); run<MOV>(d, g); run<CNDMASK>(d, g); run<LDSRD>(d, g); run<BUFLD>(d, g); run<SNOP>(d, g);
    run<FMA, true>(d, g); run<ACCRD, true>(d, g); run<MIXLO, true>(d, g);
    run<FMA, false, true>(d, g); run<MUL, false, true>(d, g); run<SPLIT3, false, true>(d, g);
    return 0;
}
