// Probe: do v_cvt_pk_f16_f32 and v_fma_mixlo_f16 round fp32 -> fp16 the same way?
//   hipcc --offload-arch=gfx950 -O3 cvt_round_probe.hip -o cvt_round_probe && ./cvt_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* x, unsigned* a, unsigned* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned ra, rb;
    const float v = x[i], one = 1.0f, zero = 0.0f;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(ra) : "v"(v));
    asm volatile("v_mov_b32 %0, 0\n\tv_fma_mixlo_f16 %0, %1, %2, %3" : "=&v"(rb) : "v"(v), "v"(one), "v"(zero));
    a[i] = ra & 0xffff; b[i] = rb & 0xffff;
}
int main() {
    const int n = 1 << 20;
    float* hx = new float[n];
    unsigned s = 1;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; unsigned bits = 0x3f000000u | (s >> 9); memcpy(&hx[i], &bits, 4); if (i & 1) hx[i] *= 1e-5f; }
    float* dx; unsigned *da, *db; hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    unsigned* ha = new unsigned[n]; unsigned* hb = new unsigned[n];
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    int diff = 0, rne_a = 0, rne_b = 0;
    for (int i = 0; i < n; ++i) {
        _Float16 r = (_Float16)hx[i]; unsigned short rb16; memcpy(&rb16, &r, 2);
        diff += ha[i] != hb[i]; rne_a += ha[i] != rb16; rne_b += hb[i] != rb16;
        if (ha[i] != hb[i] && diff < 4) printf("x=%.9g cvt_pk=%04x fma_mix=%04x host_rne=%04x\n", hx[i], ha[i], hb[i], rb16);
    }
    printf("n=%d  cvt_pk != fma_mix: %d   cvt_pk != host RNE: %d   fma_mix != host RNE: %d\n", n, diff, rne_a, rne_b);
    return 0;
}
