// Timing of the FPS kernels (dedf_graph.h) on the synthetic scene (60 % plane, 40 % cylinder): exhaustive, bucketed, bucketed + batched
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-slp-vectorize -I diffusion_edf_amd/csrc tests/probe/fps_bucket_probe.hip -o tests/probe/fps_bucket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "dedf_graph.h"
using namespace dedf;
__global__ __launch_bounds__(256) void k_poison_lds(int* sink) {      // leaves garbage in the CU's LDS (a fresh process finds zeros there)
    __shared__ int junk[38000];
    for (int i = threadIdx.x; i < 38000; i += 256) junk[i] = 0x7fc0dead + i;
    __syncthreads();
    if (junk[(threadIdx.x * 977) % 38000] == 1) *sink = 1;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const float ratio = argc > 2 ? atof(argv[2]) : 0.2f;
    const int k = (int)std::ceil(ratio * n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> U(0, 1);
    std::vector<float> x(3 * n);
    const int np = (int)std::lround(0.6 * n);
    for (int i = 0; i < n; ++i) {
        if (i < np) { x[3 * i] = -25 + 50 * U(g); x[3 * i + 1] = -25 + 50 * U(g); x[3 * i + 2] = 0; }
        else { const float th = 6.2831853f * U(g); x[3 * i] = 4 * cosf(th); x[3 * i + 1] = 4 * sinf(th); x[3 * i + 2] = 10 * U(g); }
    }
    const bool poison = argc > 3;          // third argument: fill the LDS of every CU with garbage before each launch
    float* dx; int *d0, *d1, *dsink;
    hipMalloc(&dsink, 4);
    hipMalloc(&dx, x.size() * 4); hipMalloc(&d0, k * 4); hipMalloc(&d1, k * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int which, int* out, int ks) {
        if (poison) hipLaunchKernelGGL(k_poison_lds, dim3(512), dim3(256), 0, 0, dsink);
        if (which == 0) {
            if (n <= 4096) hipLaunchKernelGGL((k_fps<16, true, 256>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
            else if (n <= 8192) hipLaunchKernelGGL((k_fps<16, true, 512>), dim3(1), dim3(512), 0, 0, dx, n, ks, 0, out);
            else hipLaunchKernelGGL((k_fps<32, true, 512>), dim3(1), dim3(512), 0, 0, dx, n, ks, 0, out);
        } else if (which == 1) {
            if (n <= 4096) hipLaunchKernelGGL((k_fps_bucketed<16>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
            else if (n <= 8192) hipLaunchKernelGGL((k_fps_bucketed<32>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
            else hipLaunchKernelGGL((k_fps_bucketed<64>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
        } else if (which == 2) {
#ifndef PROBE_NO_BATCH
            if (n <= 4096) hipLaunchKernelGGL((k_fps_bucketed<16, 256, true>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
            else if (n <= 8192) hipLaunchKernelGGL((k_fps_bucketed<32, 256, true>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
            else hipLaunchKernelGGL((k_fps_bucketed<64, 256, true>), dim3(1), dim3(256), 0, 0, dx, n, ks, 0, out);
#endif
        }
#ifdef PROBE_X8
        if (which == 3) {
            if (n <= 4096) hipLaunchKernelGGL((k_fps_bucketed<8, 512, true>), dim3(1), dim3(512), 0, 0, dx, n, ks, 0, out);
            else if (n <= 8192) hipLaunchKernelGGL((k_fps_bucketed<16, 512, true>), dim3(1), dim3(512), 0, 0, dx, n, ks, 0, out);
            else hipLaunchKernelGGL((k_fps_bucketed<32, 512, true>), dim3(1), dim3(512), 0, 0, dx, n, ks, 0, out);
        }
#endif
    };
    #ifdef PROBE_X8
    constexpr int NV = 4;
#else
    constexpr int NV = 3;
#endif
    const char* names[4] = {"plain      ", "bucketed x4", "batched    ", "batched x8 "};
    std::vector<int> h[4];
    for (int which = 0; which < NV; ++which) {
        int* out = which ? d1 : d0;
        run(which, out, k); hipDeviceSynchronize();
        for (int ks : {1, k / 8, k / 2, k}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a); run(which, out, ks); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
            }
            printf("%s n=%d samples=%d: %.3f ms (%.2f us/sample)\n", names[which], n, ks, best, best * 1e3 / ks);
        }
        h[which].resize(k);
        hipMemcpy(h[which].data(), out, k * 4, hipMemcpyDeviceToHost);
    }
    for (int which = 1; which < NV; ++which) {
        int bad = -1; for (int i = 0; i < k; ++i) if (h[0][i] != h[which][i]) { bad = i; break; }
        printf("%s against plain, first difference: %d\n", names[which], bad);
    }
    return 0;
}
