#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, float* mf) {
    const unsigned lane = threadIdx.x & 63;
    u32x2 r = __builtin_amdgcn_permlane16_swap(lane, 100 + lane, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    u32x2 s = __builtin_amdgcn_permlane32_swap(lane, 100 + lane, false, false);
    out[128 + lane] = s[0]; out[192 + lane] = s[1];
    // 16x16x4: A[i][k] = 1 if (i == 3 && k == 2) ; B[k][j] = 100*k + j  -> D[3][j] = 200 + j
    const int i = lane & 15, kk = lane >> 4;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32((i == 3 && kk == 2) ? 1.0f : 0.0f, 100.0f * kk + (lane & 15), c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) mf[lane * 4 + q] = c[q];
}
int main() {
    unsigned h[256], *d; float hm[256], *dm;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dm, sizeof(hm));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dm);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(hm, dm, sizeof(hm), hipMemcpyDeviceToHost);
    const char* nm[4] = {"swap16 r0", "swap16 r1", "swap32 r0", "swap32 r1"};
    for (int a = 0; a < 4; ++a) { printf("%s:", nm[a]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[a * 64 + l]); printf("\n"); }
    printf("mfma16 nonzero outputs (lane,reg,val):");
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) if (hm[l * 4 + q] != 0) printf(" (%d,%d,%.0f)", l, q, hm[l * 4 + q]);
    printf("\n");
    return 0;
}
