// Graph-construction primitives of the feature extractors (SURVEY §8(f) row 1 building blocks): farthest point sampling and the
// radius search, as reference diffusion_edf/connectivity.py uses them through torch_cluster (un-vendored; semantics restated in
// oracle/graph_oracle.py):
//   fps(src, ratio, random_start=False)            connectivity.py:62    -> k_fps
//   radius(x, y, r, max_num_neighbors)             connectivity.py:43    -> k_radius<false> (count) + k_radius<true> (fill)
//   radius_graph(x, r, loop=False, max_num_neigh.) connectivity.py:22    -> the same with exclude_self
// Index work: results are bit-exact against the oracle (distances are formed with explicitly rounded fp32 operations, no FMA
// contraction, ties go to the smaller index).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dedf {

constexpr int kFpsBlock = 1024;

// squared distance with the rounding sequence of the oracle: ((dx*dx + dy*dy) + dz*dz), every operation rounded to fp32
__device__ __forceinline__ float dist2_rn(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Farthest point sampling of ONE cloud by one workgroup: every thread keeps PPT points and their running minimum distance in
// registers; per sample: update, per-thread arg-max, wave butterfly, one LDS exchange between the 16 waves, ONE barrier.
// idx_out[i] = i-th selected point (selection order, first = `start`).  Ties: smaller index (numpy argmax).
template <int PPT>
__global__ __launch_bounds__(kFpsBlock) void k_fps(const float* __restrict__ x, int n, int n_samples, int start, int* __restrict__ idx_out) {
    __shared__ float s_val[2][kFpsBlock / 64];
    __shared__ int s_idx[2][kFpsBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * kFpsBlock;
        px[j] = i < n ? x[3 * i] : 0.0f; py[j] = i < n ? x[3 * i + 1] : 0.0f; pz[j] = i < n ? x[3 * i + 2] : 0.0f;
        md[j] = INFINITY;
    }
    int cur = start;
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) idx_out[s] = cur;
        const float cx = x[3 * cur], cy = x[3 * cur + 1], cz = x[3 * cur + 2];
        float best = -1.0f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = tid + j * kFpsBlock;
            if (i < n) {
                md[j] = fminf(md[j], dist2_rn(px[j], py[j], pz[j], cx, cy, cz));
                if (md[j] > best) { best = md[j]; bi = i; }          // ascending i inside a thread: strict > keeps the smaller index
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        const int buf = s & 1;
        if (lane == 0) { s_val[buf][wave] = best; s_idx[buf][wave] = bi; }
        __syncthreads();
        best = s_val[buf][0]; bi = s_idx[buf][0];
#pragma unroll
        for (int w = 1; w < kFpsBlock / 64; ++w) {
            const float ov = s_val[buf][w];
            const int oi = s_idx[buf][w];
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        cur = bi;
    }
}

// Radius search, one thread per destination point, sources streamed through LDS; neighbours in ascending source index, at most
// `cap` per destination (the first ones).  FILL = false: cnt[d];  FILL = true: edges at off[d] (exclusive scan of cnt).
constexpr int kRadChunk = 1024, kRadBlock = 256;
template <bool FILL>
__global__ __launch_bounds__(kRadBlock) void k_radius(const float* __restrict__ xs, int n_src, const float* __restrict__ xd, int n_dst, float r2, int cap,
                                                     int exclude_self, int* __restrict__ cnt, const int64_t* __restrict__ off,
                                                     int64_t* __restrict__ edge_dst, int64_t* __restrict__ edge_src) {
    __shared__ float sx[kRadChunk * 3];
    const int d = blockIdx.x * kRadBlock + threadIdx.x;
    const bool live = d < n_dst;
    const float qx = live ? xd[3 * d] : 0.0f, qy = live ? xd[3 * d + 1] : 0.0f, qz = live ? xd[3 * d + 2] : 0.0f;
    int c = 0;
    int64_t o = 0;
    if (FILL && live) o = off[d];
    for (int base = 0; base < n_src; base += kRadChunk) {
        const int m = min(kRadChunk, n_src - base);
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * m; i += kRadBlock) sx[i] = xs[3 * base + i];
        __syncthreads();
        if (!live) continue;
        for (int k = 0; k < m && c < cap; ++k) {
            if (exclude_self && base + k == d) continue;
            // the oracle forms (y - x)^2 summed in order: destination minus source
            if (dist2_rn(qx, qy, qz, sx[3 * k], sx[3 * k + 1], sx[3 * k + 2]) < r2) {
                if (FILL) { edge_dst[o + c] = d; edge_src[o + c] = base + k; }
                ++c;
            }
        }
    }
    if (!FILL && live) cnt[d] = c;
}

// exclusive scan of cnt[n] -> off[n], total -> *total  (one workgroup; n up to a few 100 k)
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ cnt, int n, int64_t* __restrict__ off, int64_t* __restrict__ total) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024, i0 = min(n, tid * per), i1 = min(n, i0 + per);
    int64_t s = 0;
    for (int i = i0; i < i1; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = part[tid] - s;
    for (int i = i0; i < i1; ++i) { off[i] = run; run += cnt[i]; }
    if (tid == 1023) *total = part[1023];
}

}  // namespace dedf
