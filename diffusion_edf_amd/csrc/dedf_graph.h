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

// ---- bucketed FPS (clouds of 1 025 .. 16 384 points) -----------------------------------------------------------------------------------
// The plain kernel above touches every point for every sample although, once a few dozen samples exist, a new sample only lowers the running
// minimum of the points NEAR it.  Here the cloud is first put into Morton order (counting sort in LDS) and cut into buckets of 128 consecutive points;
// bucket b lives in slot b / 4 of wave b % 4 (neighbouring buckets on different waves), one point pair per lane, coordinates and minima in
// registers (one wave per SIMD: 512 registers per lane).  Per bucket the wave keeps the bounding box, the largest running minimum and the
// point that holds it (lane j of the wave holds the record of slot j).  Per sample:
//   1. lane j tests its bucket: if the squared distance from the sample to the box (rounded down) is >= the bucket's largest minimum, no
//      point of the bucket can change (its distance to the sample is at least its current minimum) and the bucket is skipped -- the
//      result is therefore IDENTICAL to the exhaustive update, whatever the buckets look like;
//   2. the wave walks its active buckets: packed distance update (same rounding sequence as the oracle), DPP maximum, the owner of the
//      maximum (ties: smallest original index) leaves its coordinates in LDS;
//   3. the best bucket of the wave goes with its coordinates into a double-buffered exchange slot; ONE barrier; every thread reads the four
//      slots and knows the next sample and its coordinates.
// The loop is a serial chain run by one wave per SIMD, where a TAKEN branch costs about as much as fifteen instructions (measured: the
// first version, with a branch per rare case, spent 1 000 cycles per active bucket on 60 instructions): the hot path is written to fall through.
// Measured (tests/probe/fps_bucket_probe.hip, synthetic scene, ratio 0.2): 16 384 points 6.4 -> 4.2 ms (set-up 0.09 ms; late in the run 5 of
// the 128 buckets are active per sample and a sample costs 1.2 us: 0.4 us test + exchange, 0.1 us the wave's best bucket, the rest the
// active buckets of the busiest wave at ~0.3 us each -- a serial chain of ~70 dependent instructions per bucket on a lone wave).  At 8 000
// points it ties with the exhaustive kernel (1.97 against 2.05 ms), at 4 096 it loses (1.03 against 0.88 ms): dedf_fps uses it above 8 192.
__device__ __forceinline__ float wave_max_f32(float v) {
    auto step = [&]<int CTRL, int ROW_MASK>() {
        const int b = __float_as_int(v);
        v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xf, false)));
    };
    step.template operator()<0xB1, 0xf>(); step.template operator()<0x4E, 0xf>(); step.template operator()<0x141, 0xf>();
    step.template operator()<0x140, 0xf>(); step.template operator()<0x142, 0xa>(); step.template operator()<0x143, 0xc>();
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// maximum of a signed int over the wave, one DPP instruction per step (the running minima are >= 0 or the padding value -1: as floats
// they order like their bit patterns read as signed integers, and the integer maximum needs no NaN canonicalisation)
__device__ __forceinline__ int wave_max_i32(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
#endif
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
#endif
    return __builtin_amdgcn_readlane(v, 63);
}
// f.operator()<j>() for a wave-uniform j < N <= 32: ONE switch, so that every case leaves through the same join (a recursive if / else tree
// leaves through one join block per level, each a taken branch)
template <int N, class F> __device__ __forceinline__ void fps_dispatch(int j, F&& f) {
#define DEDF_FPS_CASE(J) case J: if constexpr (J < N) f.template operator()<J>(); break;
    switch (j) {
        DEDF_FPS_CASE(0) DEDF_FPS_CASE(1) DEDF_FPS_CASE(2) DEDF_FPS_CASE(3) DEDF_FPS_CASE(4) DEDF_FPS_CASE(5) DEDF_FPS_CASE(6) DEDF_FPS_CASE(7)
        DEDF_FPS_CASE(8) DEDF_FPS_CASE(9) DEDF_FPS_CASE(10) DEDF_FPS_CASE(11) DEDF_FPS_CASE(12) DEDF_FPS_CASE(13) DEDF_FPS_CASE(14) DEDF_FPS_CASE(15)
        DEDF_FPS_CASE(16) DEDF_FPS_CASE(17) DEDF_FPS_CASE(18) DEDF_FPS_CASE(19) DEDF_FPS_CASE(20) DEDF_FPS_CASE(21) DEDF_FPS_CASE(22) DEDF_FPS_CASE(23)
        DEDF_FPS_CASE(24) DEDF_FPS_CASE(25) DEDF_FPS_CASE(26) DEDF_FPS_CASE(27) DEDF_FPS_CASE(28) DEDF_FPS_CASE(29) DEDF_FPS_CASE(30) DEDF_FPS_CASE(31)
        default: break;
    }
#undef DEDF_FPS_CASE
}
__device__ __forceinline__ unsigned morton5(unsigned v) {      // 5 bits -> every third bit
    v &= 0x1f;
    v = (v | (v << 8)) & 0x100f; v = (v | (v << 4)) & 0x10c3; v = (v | (v << 2)) & 0x1249;
    return v;
}

#ifndef DEDF_FPS_TIE_BALLOT
#define DEDF_FPS_TIE_BALLOT 0      // (measured, round 4: the index of a unique maximum by one v_readlane instead of the second reduction -- 4.196 against 4.20 ms at 16 384 points, bit-exact, no gain: profiles/r04r_fps_time.log)
#endif
constexpr int kFpsBucketBlock = 256;
template <int PPT>      // points per thread: the cloud has at most 256 * PPT points (PPT = 16, 32, 64)
__global__ __launch_bounds__(kFpsBucketBlock) void k_fps_bucketed(const float* __restrict__ x, int n, int n_samples, int start, int* __restrict__ idx_out) {
#pragma clang fp contract(off)
    constexpr int BLOCK = kFpsBucketBlock, NW = BLOCK / 64, NP = PPT / 2, NPAD = BLOCK * PPT;
    static_assert(NP <= 32 && NPAD <= 16384, "bucket masks are 32 bits wide; original indices are kept in 16 bits");
    constexpr int kBins = 32768;
    __shared__ unsigned s_hist[kBins / 2];                  // set-up only: cell counters / offsets, two 16-bit halves per word
    __shared__ unsigned short s_perm[NPAD];                 // set-up only: original index of sorted position p
    __shared__ unsigned s_part[BLOCK];
    __shared__ float4 s_cand[NW][NP];                       // per bucket: coordinates (+ original index) of the point with the largest minimum
    __shared__ float4 s_exch[2][NW];                        // per wave: (bits of its largest minimum, coordinates of the point that holds it)
    __shared__ int s_exch_idx[2][NW];                       //           and that point's original index
    __shared__ float s_box[2][NW][3];
    __shared__ int s_out[NW][BLOCK];                        // (every wave keeps the list: no wave branches around an empty store)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- Morton order of the cloud: counting sort on a 15-bit code (5 bits per axis) ----------------------------------------------------
    // (the order INSIDE a cell is whatever the LDS atomics produce: the buckets only decide which updates are skipped, never the result)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += BLOCK)
        for (int k = 0; k < 3; ++k) { const float v = x[3 * i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
    for (int k = 0; k < 3; ++k) {
        lo[k] = -wave_max_f32(-lo[k]); hi[k] = wave_max_f32(hi[k]);
        if (lane == 0) { s_box[0][wave][k] = lo[k]; s_box[1][wave][k] = hi[k]; }
    }
    for (int i = tid; i < kBins / 2; i += BLOCK) s_hist[i] = 0u;
    __syncthreads();
    float qs[3];
    for (int k = 0; k < 3; ++k) {
        for (int w = 0; w < NW; ++w) { lo[k] = fminf(lo[k], s_box[0][w][k]); hi[k] = fmaxf(hi[k], s_box[1][w][k]); }
        const float ext = hi[k] - lo[k];
        qs[k] = ext > 0.0f ? 31.999f / ext : 0.0f;
    }
    auto cell = [&](int i) {
        unsigned q[3];
        for (int k = 0; k < 3; ++k) q[k] = (unsigned)min(31, max(0, (int)((x[3 * i + k] - lo[k]) * qs[k])));
        return morton5(q[0]) | (morton5(q[1]) << 1) | (morton5(q[2]) << 2);
    };
    // two 16-bit counters per word (n <= 16 384 < 65 536: a half never carries into its neighbour)
    for (int i = tid; i < n; i += BLOCK) { const unsigned c = cell(i); atomicAdd(&s_hist[c >> 1], 1u << (16 * (c & 1u))); }
    __syncthreads();
    {   // exclusive scan: every thread owns kBins / BLOCK consecutive cells
        constexpr int WPT = kBins / 2 / BLOCK;
        unsigned sum = 0;
        for (int i = 0; i < WPT; ++i) { const unsigned w = s_hist[tid * WPT + i]; sum += (w & 0xffffu) + (w >> 16); }
        s_part[tid] = sum;
        __syncthreads();
        unsigned run = 0;
        for (int t = 0; t < tid; ++t) run += s_part[t];
        for (int i = 0; i < WPT; ++i) {
            const unsigned w = s_hist[tid * WPT + i];
            const unsigned c0 = w & 0xffffu, c1 = w >> 16;
            s_hist[tid * WPT + i] = run | ((run + c0) << 16);
            run += c0 + c1;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BLOCK) {
        const unsigned c = cell(i), sh = 16 * (c & 1u);
        const unsigned pos = (atomicAdd(&s_hist[c >> 1], 1u << sh) >> sh) & 0xffffu;
        s_perm[pos] = (unsigned short)i;
    }
    __syncthreads();

    // ---- this thread's points: slot j of the wave is bucket 4 j + wave = sorted positions 128 (4 j + wave) + {lane, 64 + lane} -----------
    fps_f2 px[NP], py[NP], pz[NP], md[NP];
    unsigned pidx[NP];                                      // original indices of the pair, 16 bits each
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};      // lane j: box of slot j
    float bmax = -1.0f;                                     // lane j: largest running minimum of slot j (-1: no point)
    int bidx = 0;                                           // lane j: original index of the point that holds it
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p0 = 128 * (NW * j + wave) + lane, p1 = p0 + 64;
        const bool v0 = p0 < n, v1 = p1 < n;
        // padding slots read the last sorted point and their minimum is pinned at -1 below.  NOT a guarded load: with `v0 ? s_perm[p0] : 0` the
        // -fno-slp-vectorize build (ROCm 7.2 hipcc, 42 SGPRs spilled to VGPR lanes around the masked loads) left every lane whose first point
        // is padding disabled for the rest of the kernel -- its points were never selected and its share of the output never stored
        // (tests/probe/fps_bucket_probe.hip reproduces it; the default flags and the 16 384-point case were fine)
        const int i0 = (int)s_perm[min(p0, n - 1)], i1 = (int)s_perm[min(p1, n - 1)];
        px[j] = fps_f2{x[3 * i0], x[3 * i1]}; py[j] = fps_f2{x[3 * i0 + 1], x[3 * i1 + 1]}; pz[j] = fps_f2{x[3 * i0 + 2], x[3 * i1 + 2]};
        md[j] = fps_f2{v0 ? INFINITY : -1.0f, v1 ? INFINITY : -1.0f};
        pidx[j] = (unsigned)i0 | ((unsigned)i1 << 16);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p0 = 128 * (NW * j + wave) + lane;
        const bool v0 = p0 < n, v1 = p0 + 64 < n;
        const float c[3][2] = {{px[j].x, px[j].y}, {py[j].x, py[j].y}, {pz[j].x, pz[j].y}};
        for (int k = 0; k < 3; ++k) {
            const float mn = -wave_max_f32(-fminf(v0 ? c[k][0] : INFINITY, v1 ? c[k][1] : INFINITY));
            const float mx = wave_max_f32(fmaxf(v0 ? c[k][0] : -INFINITY, v1 ? c[k][1] : -INFINITY));
            if (lane == j) { blo[k] = mn; bhi[k] = mx; }
        }
        const bool any_valid = __any(v0) != 0;              // (every lane takes part: the bucket's first half fills first)
        if (lane == j && any_valid) bmax = INFINITY;
    }

    int cur = start;
    float cx = x[3 * cur], cy = x[3 * cur + 1], cz = x[3 * cur + 2];
    // the wave's best bucket, kept while no bucket of the wave changes
    int wmd = __float_as_int(-1.0f), widx = 0x7fffffff;
    float wx = 0.0f, wy = 0.0f, wz = 0.0f;
    for (int s = 0; s < n_samples; ++s) {
        if (lane == 0) s_out[wave][s & (BLOCK - 1)] = cur;
#if defined(DEDF_FPS_STATS)
        const long long tk0 = __builtin_readcyclecounter();
#endif
        // 1. which buckets of this wave can change
        const float ddx = fmaxf(fmaxf(blo[0] - cx, cx - bhi[0]), 0.0f), ddy = fmaxf(fmaxf(blo[1] - cy, cy - bhi[1]), 0.0f),
                    ddz = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.0f);
        const float lb2 = ((ddx * ddx + ddy * ddy) + ddz * ddz) * (1.0f - 4e-6f);      // below every distance the update would compute
        unsigned mask = (unsigned)__ballot(lane < NP && !(lb2 >= bmax));
        // 2. active buckets
#if defined(DEDF_FPS_STATS)
        if (lane == 0) { atomicAdd(&g_fps_stats[0], (unsigned long long)__builtin_popcount(mask)); atomicAdd(&g_fps_stats[1], mask != 0u ? 1ull : 0ull); }
        const long long tk05 = __builtin_readcyclecounter();
        if (tid == 64) { atomicAdd(&g_fps_stats[7], (unsigned long long)(tk05 - tk0)); }
#endif
#if defined(DEDF_FPS_SKIP) && DEDF_FPS_SKIP >= 1
        if (s > 0) mask = 0u;          // timing experiment (wrong results): no bucket updates after the first sample
#endif
        while (mask) {
            const int ja = __builtin_ctz(mask);
            mask &= mask - 1u;
#if defined(DEDF_FPS_SKIP) && DEDF_FPS_SKIP == 3
            fps_dispatch<1>(ja & 0, [&]<int J>() {      // timing experiment (wrong results): every active bucket runs slot 0's code
#else
            fps_dispatch<NP>(ja, [&]<int J>() {
#endif
                // (opaque copy of the sample: hipcc otherwise hoists the distance arithmetic of ALL slots out of the dispatch and computes it
                // for every sample -- the exhaustive update again)
                float ox = cx, oy = cy, oz = cz;
                asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz));
                const fps_f2 c_x = fps_f2{ox, ox}, c_y = fps_f2{oy, oy}, c_z = fps_f2{oz, oz};
                const fps_f2 dx = px[J] - c_x, dy = py[J] - c_y, dz = pz[J] - c_z;
                const fps_f2 d2 = (dx * dx + dy * dy) + dz * dz;
                fps_f2 m = md[J];
                m.x = fminf(m.x, d2.x); m.y = fminf(m.y, d2.y);
                md[J] = m;
                const int b0 = __float_as_int(m.x), b1 = __float_as_int(m.y);
                const int i0 = (int)(pidx[J] & 0xffffu), i1 = (int)(pidx[J] >> 16);
                // the lane's better point (larger minimum, then smaller original index) and its coordinates: independent of the reduction
                const bool sel1 = b1 > b0 || (b1 == b0 && i1 < i0);
                const int lb = sel1 ? b1 : b0, li = sel1 ? i1 : i0;
                const float4 lc = float4{sel1 ? px[J].y : px[J].x, sel1 ? py[J].y : py[J].x, sel1 ? pz[J].y : pz[J].x, 0.0f};
                const int wmi = wave_max_i32(lb);                                         // wave-uniform
                const float wm = __int_as_float(wmi);
                // (a taken branch costs a lone wave about as much as fifteen instructions: ties are resolved by a second reduction, always)
#if DEDF_FPS_TIE_BALLOT
                // ties are rare: when exactly one lane holds the maximum (the fall-through path) its index comes from one v_readlane instead of a
                // second six-step reduction; exact ties take the reduction (round 4)
                const unsigned long long tied = __ballot(lb == wmi);
                int win;
                if (__builtin_expect(__builtin_popcountll(tied) == 1, 1)) win = __builtin_amdgcn_readlane(li, __builtin_ctzll(tied));
                else win = wave_min_i32_dpp(lb == wmi ? li : 0x7fffffff);
#else
                const int win = wave_min_i32_dpp(lb == wmi ? li : 0x7fffffff);
#endif
                if (li == win && lb == wmi && wmi >= 0) s_cand[wave][J] = lc;
                if (lane == J) { bmax = wm; bidx = win; }
            });
        }
#if defined(DEDF_FPS_STATS)
        const long long tk1 = __builtin_readcyclecounter();
#endif
        // 3. best bucket of the wave: largest minimum, ties to the smaller original index (recomputed every sample: cheaper than branching
        //    around it, and a wave without work is not the one the others wait for)
#if defined(DEDF_FPS_SKIP) && DEDF_FPS_SKIP >= 2
        if (s == 0)
#endif
        {
            const int kb = lane < NP ? __float_as_int(bmax) : -1;
            const int best = wave_max_i32(kb);
#if DEDF_FPS_TIE_BALLOT
            const unsigned long long tiedb = __ballot(kb == best);
            int bi, jb;
            if (__builtin_expect(__builtin_popcountll(tiedb) == 1, 1)) { jb = __builtin_ctzll(tiedb); bi = __builtin_amdgcn_readlane(bidx, jb); }
            else { bi = wave_min_i32_dpp(kb == best ? bidx : 0x7fffffff); jb = __builtin_ctzll(__ballot(kb == best && bidx == bi)); }
#else
            const int bi = wave_min_i32_dpp(kb == best ? bidx : 0x7fffffff);
            const int jb = __builtin_ctzll(__ballot(kb == best && bidx == bi));
#endif
            wmd = best; widx = bi;
            const float4 c = s_cand[wave][jb < NP ? jb : 0];
            wx = c.x; wy = c.y; wz = c.z;
        }
        const int buf = s & 1;
        if (lane == 0) { s_exch[buf][wave] = float4{__int_as_float(wmd), wx, wy, wz}; s_exch_idx[buf][wave] = widx; }
#if defined(DEDF_FPS_STATS)
        const long long tk2 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#if defined(DEDF_FPS_STATS)
        const long long tk3 = __builtin_readcyclecounter();
#endif
        if (__builtin_expect((s & (BLOCK - 1)) == BLOCK - 1 || s == n_samples - 1, 0)) {
            const int base = s & ~(BLOCK - 1);
            if (base + tid <= s) idx_out[base + tid] = s_out[0][tid];
        }
        // all four slots are read before anything is compared (left to itself hipcc reads them one by one behind branches: four LDS
        // round trips in a row), then a two-level knock-out without branches
        static_assert(NW == 4, "the knock-out below is written for four waves");
        float4 e[NW]; int ei[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { e[w] = s_exch[buf][w]; ei[w] = s_exch_idx[buf][w]; }
        asm volatile("" : "+v"(e[0].x), "+v"(e[1].x), "+v"(e[2].x), "+v"(e[3].x), "+v"(ei[0]), "+v"(ei[1]), "+v"(ei[2]), "+v"(ei[3]));
        auto better = [](const float4& a, int ai, const float4& b, int bi) {      // is b ahead of a
            const int am = __float_as_int(a.x), bm = __float_as_int(b.x);
            return bm > am || (bm == am && bi < ai);
        };
        const bool s01 = better(e[0], ei[0], e[1], ei[1]), s23 = better(e[2], ei[2], e[3], ei[3]);
        const float4 a01 = float4{s01 ? e[1].x : e[0].x, s01 ? e[1].y : e[0].y, s01 ? e[1].z : e[0].z, s01 ? e[1].w : e[0].w};
        const float4 a23 = float4{s23 ? e[3].x : e[2].x, s23 ? e[3].y : e[2].y, s23 ? e[3].z : e[2].z, s23 ? e[3].w : e[2].w};
        const int i01 = s01 ? ei[1] : ei[0], i23 = s23 ? ei[3] : ei[2];
        const bool sf = better(a01, i01, a23, i23);
        cur = sf ? i23 : i01;
        cx = sf ? a23.y : a01.y; cy = sf ? a23.z : a01.z; cz = sf ? a23.w : a01.w;
#if defined(DEDF_FPS_STATS)
        if (tid == 64) {      // wave 1
            const long long tk4 = __builtin_readcyclecounter();
            atomicAdd(&g_fps_stats[3], (unsigned long long)(tk1 - tk0)); atomicAdd(&g_fps_stats[4], (unsigned long long)(tk2 - tk1));
            atomicAdd(&g_fps_stats[5], (unsigned long long)(tk3 - tk2)); atomicAdd(&g_fps_stats[6], (unsigned long long)(tk4 - tk3));
        }
#endif
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
