// GPU probe: (1) v_permlane16_swap turns two registers of the 32x32 "edge column / row half" layout into the B operands of
// two v_mfma_f32_16x16x4_f32 sub-tiles (edges 0-15 | 16-31); (2) the 16x16 result maps back with swap16 + swap32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifdef ASMSWAP
// inline-asm forms: 2 wait states between a VALU write of an operand and the swap (MI355X guide T21), and the consumers
// after the statement are fenced by the compiler's boundary pad + an explicit s_nop
__device__ inline void swap16(float& x, float& y) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y)); }
__device__ inline void swap32(float& x, float& y) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y)); }
#else
__device__ inline void swap16(float& x, float& y) {   // x' = [x0,y0,x2,y2], y' = [x1,y1,x3,y3]  (16-lane rows)
    u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    x = __builtin_bit_cast(float, r[0]); y = __builtin_bit_cast(float, r[1]);
}
__device__ inline void swap32(float& x, float& y) {   // x' = [x.lo, y.lo], y' = [x.hi, y.hi]
    u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    x = __builtin_bit_cast(float, r[0]); y = __builtin_bit_cast(float, r[1]);
}
#endif
// act[k][edge]: K = 8 rows (one group: regs j=0..3, rows j + 4*hi), 32 edges; W[16][8]; out[16][32]
__global__ void k(const float* act, const float* W, float* out, float* out2) {
    const int lane = threadIdx.x & 63, e = lane & 31, hi = lane >> 5;
    float b[4];
    for (int j = 0; j < 4; ++j) b[j] = act[(j + 4 * hi) * 32 + e];       // 32x32-layout K-step registers
    f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
    for (int p = 0; p < 2; ++p) {
        float x = b[2 * p], y = b[2 * p + 1];
        swap16(x, y);                                                      // x: sub-tile A operand, y: sub-tile B operand
#ifdef NOPS
        asm volatile("s_nop 7" : "+v"(x), "+v"(y));
#endif
        const int slot = lane >> 4, o = lane & 15;
        const int kk = (slot & 1) + 2 * p + 4 * (slot >> 1);              // slot0:(hi0,2p) slot1:(hi0,2p+1) slot2:(hi1,2p) slot3:(hi1,2p+1)
        const float a = W[o * 8 + kk];
        accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a, y, accB, 0, 0, 0);
    }
    // direct store from the 16x16 layout: lane (g, e') reg q -> channel 4g+q of edge e' (+16 for sub-tile B)
    for (int q = 0; q < 4; ++q) {
        out2[(4 * (lane >> 4) + q) * 32 + (lane & 15)] = accA[q];
        out2[(4 * (lane >> 4) + q) * 32 + 16 + (lane & 15)] = accB[q];
    }
    // back to the 32x32 layout: T[r], r = 0..7 holds channel (r&3) + 8*(r>>2) + 4*hi of edge e
    float T[8];
    for (int q = 0; q < 4; ++q) {
        float x = accA[q], y = accB[q];
        swap16(x, y);
        swap32(x, y);
        T[q] = x; T[4 + q] = y;
    }
    for (int r = 0; r < 8; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + e] = T[r];
}
int main() {
    float act[8 * 32], W[16 * 8], out[16 * 32], out2[16 * 32], *da, *dw, *dout, *dout2;
    for (int i = 0; i < 256; ++i) act[i] = sinf(0.37f * i) + 0.01f * i;
    for (int i = 0; i < 128; ++i) W[i] = cosf(0.91f * i) - 0.003f * i;
    (void)hipMalloc(&da, sizeof(act)); (void)hipMalloc(&dw, sizeof(W)); (void)hipMalloc(&dout, sizeof(out)); (void)hipMalloc(&dout2, sizeof(out));
    (void)hipMemcpy(da, act, sizeof(act), hipMemcpyHostToDevice); (void)hipMemcpy(dw, W, sizeof(W), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dw, dout, dout2);
    (void)hipMemcpy(out2, dout2, sizeof(out2), hipMemcpyDeviceToHost);
    (void)hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    double err = 0, err2 = 0;
    for (int o = 0; o < 16; ++o) for (int e = 0; e < 32; ++e) {
        double r = 0; for (int kk = 0; kk < 8; ++kk) r += (double)W[o * 8 + kk] * act[kk * 32 + e];
        err = fmax(err, fabs(r - out[o * 32 + e])); err2 = fmax(err2, fabs(r - out2[o * 32 + e]));
        if (o < 2 && e < 3) printf("o %d e %d ref %.4f direct %.4f converted %.4f\n", o, e, r, out2[o * 32 + e], out[o * 32 + e]);
    }
    printf("direct-store err %.3e\n", err2);
    for (int o = 0; o < 16; ++o) { for (int e = 0; e < 32; ++e) { double r = 0; for (int kk = 0; kk < 8; ++kk) r += (double)W[o * 8 + kk] * act[kk * 32 + e]; printf("%c", fabs(r - out2[o * 32 + e]) < 1e-4 ? '.' : 'X'); } printf("\n"); }
    printf("mfma16 + permlane swap probe: max err %.3e -> %s\n", err, err < 1e-5 ? "OK" : "FAIL");
    return err < 1e-5 ? 0 : 1;
}
