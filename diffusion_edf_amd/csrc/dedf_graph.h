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

// Farthest point sampling of ONE cloud by one workgroup of BLOCK threads: every thread keeps PPT points and their running minimum distance
// in registers; per sample: update, per-thread arg-max, wave butterfly, one LDS exchange between the waves.
// The sample loop is a serial chain of n_samples steps on ONE CU.  A step costs a fixed part that every wave issues (butterfly, exchange,
// barriers: 0.43 / 0.79 / 1.95 us with 4 / 8 / 16 waves) plus the per-point update (0.04 us per point of a thread while at most two waves
// share a SIMD), so dedf_fps picks 256 threads up to 4 096 points and 512 up to 16 384; the step itself is written for instruction count:
//   * two points per instruction with the packed fp32 ops (v_pk_add/mul_f32): 8 packed operations per PAIR for the distance, rounded
//     exactly like the oracle (fp contraction off: no FMA);
//   * the arg-max travels as ONE 64-bit key (distance bits << 32 | 0x7fffffff - index; distances are >= 0, so the unsigned order of
//     the key is "larger distance, then smaller index" = numpy argmax);
//   * the winner's coordinates come back through LDS from the thread that owns the point (no dependent global load), and the
//     selected indices are collected in LDS and written out once per 1024 samples (a barrier waits for outstanding global stores).
// idx_out[i] = i-th selected point (selection order, first = `start`).
// KEEP = false (clouds above 16 k points, 1 024 threads): only the minimum distances stay in registers, the coordinates are re-read
// (coalesced, from L2) every sample — 64 points per thread would not fit the 128 VGPRs a thread of a 1024-thread workgroup has.
typedef float fps_f2 __attribute__((ext_vector_type(2)));

// max of a 64-bit key over the wave on the DPP path (no LDS crossbar): xor 1, xor 2, half-mirror and mirror inside every row of 16 lanes,
// then the row results are passed on with row_bcast:15 / row_bcast:31; lane 63 ends up with the maximum, which is broadcast as a scalar.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long key) {
    auto step = [&]<int CTRL, int ROW_MASK>() {
        const int lo = (int)(unsigned)key, hi = (int)(unsigned)(key >> 32);
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
        const unsigned long long ok = ((unsigned long long)ohi << 32) | olo;
        key = ok > key ? ok : key;
    };
    step.template operator()<0xB1, 0xf>();      // quad_perm [1,0,3,2]
    step.template operator()<0x4E, 0xf>();      // quad_perm [2,3,0,1]
    step.template operator()<0x141, 0xf>();     // row_half_mirror
    step.template operator()<0x140, 0xf>();     // row_mirror
    step.template operator()<0x142, 0xa>();     // row_bcast:15 -> rows 1, 3
    step.template operator()<0x143, 0xc>();     // row_bcast:31 -> rows 2, 3
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

template <int PPT, bool KEEP = true, int BLOCK = kFpsBlock>
__global__ __launch_bounds__(BLOCK) void k_fps(const float* __restrict__ x, int n, int n_samples, int start, int* __restrict__ idx_out) {
#pragma clang fp contract(off)
    static_assert(PPT % 2 == 0, "points are processed in pairs");
    constexpr int NW = BLOCK / 64, NP = PPT / 2;
    __shared__ unsigned long long s_key[2][NW];
    __shared__ float s_xyz[2][4];
    __shared__ int s_out[BLOCK];                            // selected indices, flushed every BLOCK samples
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fps_f2 px[KEEP ? NP : 1], py[KEEP ? NP : 1], pz[KEEP ? NP : 1], md[NP];
    auto ld = [&](int i, int k) { return x[3 * min(i, n - 1) + k]; };            // padding slots read the last point; their md is pinned at -1
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int i0 = tid + (2 * j) * BLOCK, i1 = i0 + BLOCK;
        if constexpr (KEEP) { px[j] = fps_f2{ld(i0, 0), ld(i1, 0)}; py[j] = fps_f2{ld(i0, 1), ld(i1, 1)}; pz[j] = fps_f2{ld(i0, 2), ld(i1, 2)}; }
        md[j] = fps_f2{i0 < n ? INFINITY : -1.0f, i1 < n ? INFINITY : -1.0f};       // -1: below every real distance, stays -1 under min
    }
    int cur = start;
    float cx = x[3 * cur], cy = x[3 * cur + 1], cz = x[3 * cur + 2];
    for (int s = 0; s < n_samples; ++s) {
        if (tid == 0) s_out[s & (BLOCK - 1)] = cur;         // (a global store here would be waited for at every barrier)
        const fps_f2 c_x = fps_f2{cx, cx}, c_y = fps_f2{cy, cy}, c_z = fps_f2{cz, cz};
        int tid_s = tid;
        if constexpr (!KEEP) asm volatile("" : "+v"(tid_s));    // addresses are formed per sample: 64 loop-invariant offsets would not fit the registers
        float best = -1.0f;
        int bj = -1;                                                                   // slot (2 * pair + half) of the thread's arg-max
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            fps_f2 qx, qy, qz;
            if constexpr (KEEP) { qx = px[j]; qy = py[j]; qz = pz[j]; }
            else {
                const int i0 = tid_s + (2 * j) * BLOCK, i1 = i0 + BLOCK;
                qx = fps_f2{ld(i0, 0), ld(i1, 0)}; qy = fps_f2{ld(i0, 1), ld(i1, 1)}; qz = fps_f2{ld(i0, 2), ld(i1, 2)};
            }
            const fps_f2 dx = qx - c_x, dy = qy - c_y, dz = qz - c_z;
            const fps_f2 d2 = (dx * dx + dy * dy) + dz * dz;
            fps_f2 m = md[j];
            m.x = fminf(m.x, d2.x); m.y = fminf(m.y, d2.y);
            md[j] = m;
            if (m.x > best) { best = m.x; bj = 2 * j; }                               // ascending index inside a thread: strict > keeps the smaller one
            if (m.y > best) { best = m.y; bj = 2 * j + 1; }
            if constexpr (!KEEP) __builtin_amdgcn_sched_barrier(0);     // keep the re-loads of later pairs from piling up in registers
        }
        const int bi = tid + bj * BLOCK;
        unsigned long long key = bj < 0 ? 0ull : ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(0x7fffffff - bi);
        key = wave_max_u64(key);                                                      // wave-uniform
        const int buf = s & 1;
        if (lane == 0) s_key[buf][wave] = key;
        __syncthreads();
        if ((s & (BLOCK - 1)) == BLOCK - 1 || s == n_samples - 1) {
            const int base = s & ~(BLOCK - 1);
            if (base + tid <= s) idx_out[base + tid] = s_out[tid];
        }
        key = s_key[buf][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const unsigned long long ok = s_key[buf][w];
            key = ok > key ? ok : key;
        }
        cur = 0x7fffffff - (int)(unsigned)(key & 0xffffffffull);
        if constexpr (KEEP) {
            if (tid == (cur & (BLOCK - 1))) {                                      // the owner publishes the coordinates
                const int slot = cur / BLOCK;
                float ox = 0.0f, oy = 0.0f, oz = 0.0f;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    if (slot == 2 * j) { ox = px[j].x; oy = py[j].x; oz = pz[j].x; }
                    if (slot == 2 * j + 1) { ox = px[j].y; oy = py[j].y; oz = pz[j].y; }
                }
                s_xyz[buf][0] = ox; s_xyz[buf][1] = oy; s_xyz[buf][2] = oz;
            }
            __syncthreads();
            cx = s_xyz[buf][0]; cy = s_xyz[buf][1]; cz = s_xyz[buf][2];
        } else {
            cx = x[3 * cur]; cy = x[3 * cur + 1]; cz = x[3 * cur + 2];
        }
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
