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
// KEEP = false (clouds above 16 k points): only the minimum distances stay in registers, the coordinates are re-read (coalesced, from
// L2) every sample — 64 points per thread would not fit the 128 VGPRs a thread of a 1024-thread workgroup has.
template <int PPT, bool KEEP = true>
__global__ __launch_bounds__(kFpsBlock) void k_fps(const float* __restrict__ x, int n, int n_samples, int start, int* __restrict__ idx_out) {
    __shared__ float s_val[2][kFpsBlock / 64];
    __shared__ int s_idx[2][kFpsBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float px[KEEP ? PPT : 1], py[KEEP ? PPT : 1], pz[KEEP ? PPT : 1], md[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * kFpsBlock;
        if constexpr (KEEP) { px[j] = i < n ? x[3 * i] : 0.0f; py[j] = i < n ? x[3 * i + 1] : 0.0f; pz[j] = i < n ? x[3 * i + 2] : 0.0f; }
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
                float qx, qy, qz;
                if constexpr (KEEP) { qx = px[j]; qy = py[j]; qz = pz[j]; }
                else { qx = x[3 * i]; qy = x[3 * i + 1]; qz = x[3 * i + 2]; }
                md[j] = fminf(md[j], dist2_rn(qx, qy, qz, cx, cy, cz));
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

// Radius search, one WAVE per destination point: the 64 lanes test 64 consecutive sources per step, a ballot counts them and
// orders the hits (ascending source index for free); at most `cap` neighbours per destination (the first ones).
// FILL = false: cnt[d];  FILL = true: edges at off[d] (exclusive scan of cnt).
constexpr int kRadBlock = 256;
template <bool FILL>
__global__ __launch_bounds__(kRadBlock) void k_radius(const float* __restrict__ xs, int n_src, const float* __restrict__ xd, int n_dst, float r2, int cap,
                                                     int exclude_self, int* __restrict__ cnt, const int64_t* __restrict__ off,
                                                     int64_t* __restrict__ edge_dst, int64_t* __restrict__ edge_src) {
    const int lane = threadIdx.x & 63;
    const int d = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * kRadBlock + threadIdx.x) >> 6));
    if (d >= n_dst) return;
    const float qx = xd[3 * d], qy = xd[3 * d + 1], qz = xd[3 * d + 2];
    int c = 0;
    int64_t o = 0;
    if (FILL) o = off[d];
    for (int base = 0; base < n_src && c < cap; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n_src && !(exclude_self && k == d))
            hit = dist2_rn(qx, qy, qz, xs[3 * k], xs[3 * k + 1], xs[3 * k + 2]) < r2;       // the oracle forms (y - x)^2: destination minus source
        const unsigned long long m = __ballot(hit);
        if (FILL && hit) {
            const int pos = c + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) { edge_dst[o + pos] = d; edge_src[o + pos] = k; }
        }
        c += __popcll(m);
    }
    if (!FILL && lane == 0) cnt[d] = min(c, cap);
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
