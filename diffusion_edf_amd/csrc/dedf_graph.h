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
// points it ties with the exhaustive kernel (1.97 against 2.05 ms), at 4 096 it loses (1.03 against 0.88 ms).
// BATCH = true (round 6, what dedf_fps launches): the ~1 us round trip through the workgroup is paid once per BATCH of up to 64 samples
// instead of once per sample -- see "batches" inside.  16 384 points: 4.2 -> 1.5 ms, 8 192: 2.1 -> 0.69, 3 277: 0.61 -> 0.28 ms (8 waves,
// kernel time), ahead of the exhaustive kernel from ~1 100 points on (profiles/r06p_fps_*.log, r06r_*); still bit-identical to the exhaustive arg-max.
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

constexpr int kFpsBucketBlock = 256;
// the better of the exchange slots [LO, LO + N): larger minimum, then smaller original index (straight-line selects)
struct FpsSlot { float4 e; int i; };
template <int LO, int N, int NW> __device__ __forceinline__ FpsSlot fps_knock_out(const float4 (&e)[NW], const int (&ei)[NW]) {
    if constexpr (N == 1) return FpsSlot{e[LO], ei[LO]};
    else {
        const FpsSlot a = fps_knock_out<LO, N / 2, NW>(e, ei), b = fps_knock_out<LO + N / 2, N / 2, NW>(e, ei);
        const int am = __float_as_int(a.e.x), bm = __float_as_int(b.e.x);
        const bool sb = bm > am || (bm == am && b.i < a.i);
        return FpsSlot{float4{sb ? b.e.x : a.e.x, sb ? b.e.y : a.e.y, sb ? b.e.z : a.e.z, sb ? b.e.w : a.e.w}, sb ? b.i : a.i};
    }
}
constexpr int kFpsBatchStart = 16;         // samples taken one by one before the first batch (early samples lower every running minimum)
constexpr int kFpsCandPerLane = 4;          // candidate list of a batch: 64 lanes x this many points
constexpr int kFpsBatchedFrom = 1024;       // dedf_fps: clouds above this many points take the bucketed + batched kernel (measured, tests/probe/fps_time.py: 0.167 against 0.227 ms at 1 100 points, a tie at 656)
template <int PPT, int BLOCK = kFpsBucketBlock, bool BATCH = false>      // points per thread: the cloud has at most BLOCK * PPT points
__global__ __launch_bounds__(BLOCK) void k_fps_bucketed(const float* __restrict__ x, int n, int n_samples, int start, int* __restrict__ idx_out) {
#pragma clang fp contract(off)
    constexpr int NW = BLOCK / 64, NP = PPT / 2, NPAD = BLOCK * PPT;
    static_assert(NP <= 32 && NPAD <= 16384, "bucket masks are 32 bits wide; original indices are kept in 16 bits");
    constexpr int kBins = 32768;
    __shared__ __align__(16) unsigned s_hist[kBins / 2];    // set-up: cell counters / offsets, two 16-bit halves per word; afterwards the candidate list of a batch
    __shared__ int s_cnt[4];                                // candidate counters, one per collection attempt (rotating)
    __shared__ float4 s_batch[64];                          // 8 waves: the batch drawn by wave 0 for the waves that do not draw
    __shared__ int s_batch_m;
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
    if (tid < 4) s_cnt[tid] = 0;
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
        // (opaque: the pairs are what stays in registers -- otherwise hipcc also keeps the loaded (x, y, z) triples alive for every scalar use)
        asm volatile("" : "+v"(px[j]), "+v"(py[j]), "+v"(pz[j]));
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
    // BATCH (round 6): see "batches" below; these are wave-uniform
    [[maybe_unused]] int curm = __float_as_int(INFINITY);            // bits of the running minimum of `cur` (the largest one)
    [[maybe_unused]] int single_until = BATCH ? min(n_samples, kFpsBatchStart) : n_samples, flushed = 0, attempt = 0, it = 0;
    [[maybe_unused]] float frac = 0.9f;
    for (int s = 0; s < n_samples;) {
        int n_adv = 1;                                      // samples this iteration settles
        bool single = true;
        if constexpr (BATCH) {
            // ---- batches -------------------------------------------------------------------------------------------------------------------
            // Once a few dozen samples exist the running minima form a plateau: many points within a few per cent of the largest one, far
            // apart from each other.  Let tau <= the largest minimum and C = {points whose minimum is >= tau}.  Every other point is below tau
            // and can only fall, so AS LONG AS the best point of C (minima of C kept exact against the samples drawn meanwhile) is >= tau, it is
            // the next sample of the exhaustive algorithm -- found by ONE wave inside its registers (<= 256 candidates, 4 per lane), no barrier,
            // no LDS: ~0.2 us per sample against ~1.2 us for a sample that goes round the workgroup.  The samples of a batch (<= 64: lane i keeps
            // sample i) are then applied to the buckets in one pass (a bucket is touched once per BATCH: one pair of reductions for all its
            // hits), the bucket maxima are exact again and the exchange below names the first sample of the next batch.  With 4 waves every
            // wave draws the same batch redundantly (same instructions, same data: same result: nothing to broadcast); with 8 only waves 0-3
            // do, one per SIMD at full speed, and the other four read the batch from LDS behind a barrier.
            // tau = frac * largest minimum; frac rises when the candidates overflow the list and falls when a batch ends because the list ran dry.
            single = s < single_until;
            constexpr int C = kFpsCandPerLane, CAP = 64 * C;
            // (component by component: a float4 per candidate would make hipcc keep a second, (x, y, z, -)-shaped copy of every point in registers)
            float* const s_cx = reinterpret_cast<float*>(s_hist), * const s_cy = s_cx + CAP, * const s_cz = s_cy + CAP, * const s_cm = s_cz + CAP;
            int* const s_ci = reinterpret_cast<int*>(s_cm + CAP);
            float tau = 0.0f;
            int cnt = 0;
            if (!single) {
                bool ok = false;
                const float M = __int_as_float(__builtin_amdgcn_readfirstlane(curm));        // (read from LDS: uniform, but not to the compiler)
                for (int t = 0; t < 3 && !ok; ++t) {
                    tau = M * frac;
                    if (!(tau > 0.0f) || !(tau < INFINITY)) break;
                    int* const c_now = &s_cnt[attempt & 3];
                    if (tid == 0) s_cnt[(attempt + 2) & 3] = 0;          // (last read two barriers ago)
                    ++attempt;
                    int nw = 0;
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                        nw += __builtin_popcountll(__ballot(md[j].x >= tau)) + __builtin_popcountll(__ballot(md[j].y >= tau));
                    int base = 0;
                    if (lane == 0) base = atomicAdd(c_now, nw);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + nw <= CAP) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            const bool c0 = md[j].x >= tau, c1 = md[j].y >= tau;
                            const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1);
                            if (b0 | b1) {
                                const int n0 = __builtin_popcountll(b0);
                                const int p0 = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
                                const int p1 = base + n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
                                unsigned pj = pidx[j];
                                asm volatile("" : "+v"(pj));          // (else the 2 x NP unpacked indices are hoisted out of the sample loop into registers)
                                if (c0) { s_cx[p0] = px[j].x; s_cy[p0] = py[j].x; s_cz[p0] = pz[j].x; s_cm[p0] = md[j].x; s_ci[p0] = (int)(pj & 0xffffu); }
                                if (c1) { s_cx[p1] = px[j].y; s_cy[p1] = py[j].y; s_cz[p1] = pz[j].y; s_cm[p1] = md[j].y; s_ci[p1] = (int)(pj >> 16); }
                                base += n0 + __builtin_popcountll(b1);
                            }
                        }
                    }
                    __syncthreads();
                    cnt = __builtin_amdgcn_readfirstlane(*c_now);
                    if (cnt <= CAP) ok = true;
                    else frac = 0.5f * (1.0f + frac);
                }
                if (!ok) { single_until = min(n_samples, s + 4); single = true; }          // (ties / duplicates / the first samples: one by one for a while)
            }
            if (!single) {
                if (flushed < s) {                              // samples still waiting in the ring
                    if (flushed + tid < s) idx_out[flushed + tid] = s_out[0][(flushed + tid) & (BLOCK - 1)];
                }
                const int m_max = min(64, n_samples - s), tau_bits = __builtin_amdgcn_readfirstlane(__float_as_int(tau));
                float sx = 0.0f, sy = 0.0f, sz = 0.0f;          // lane i: sample i of the batch
                int si = 0;
                int m = 0;
                bool dry = false;
                // with 8 waves two share a SIMD: only waves 0-3 draw (one per SIMD, at full speed), the others get the batch through LDS
                constexpr bool SPLIT = NW == 8;
                const bool draws = !SPLIT || __builtin_amdgcn_readfirstlane(wave) < 4;
                if (draws) {
                    // the candidates, C per lane as C / 2 pairs (minimum -1: none)
                    static_assert(C % 2 == 0, "candidates are processed in pairs");
                    fps_f2 qx[C / 2], qy[C / 2], qz[C / 2], qm[C / 2];
                    int qi[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int k = lane + 64 * c;
                        const bool valid = k < cnt;
                        const int kk = valid ? k : 0;
                        const float vx = s_cx[kk], vy = s_cy[kk], vz = s_cz[kk], vm = valid ? s_cm[kk] : -1.0f;
                        if (c & 1) { qx[c / 2].y = vx; qy[c / 2].y = vy; qz[c / 2].y = vz; qm[c / 2].y = vm; }
                        else { qx[c / 2].x = vx; qy[c / 2].x = vy; qz[c / 2].x = vz; qm[c / 2].x = vm; }
                        qi[c] = valid ? s_ci[kk] : 0x7fffffff;
                    }
                    for (; m < m_max; ++m) {
                        // the lane's best candidate (larger minimum, then smaller index; minima are >= 0 or -1: they order like their bits)
                        int bm = __float_as_int(qm[0].x), bi = qi[0];
                        float bx = qx[0].x, by = qy[0].x, bz = qz[0].x;
#pragma unroll
                        for (int c = 1; c < C; ++c) {
                            const int cm = __float_as_int((c & 1) ? qm[c / 2].y : qm[c / 2].x);
                            const bool t = (cm > bm) | ((cm == bm) & (qi[c] < bi));          // (no short circuit: that would be branches)
                            bm = t ? cm : bm; bi = t ? qi[c] : bi;
                            bx = t ? ((c & 1) ? qx[c / 2].y : qx[c / 2].x) : bx; by = t ? ((c & 1) ? qy[c / 2].y : qy[c / 2].x) : by;
                            bz = t ? ((c & 1) ? qz[c / 2].y : qz[c / 2].x) : bz;
                        }
                        const int wmi = wave_max_i32(bm);
                        if (wmi < tau_bits) { dry = true; break; }
                        // the lane that holds it: one lane unless two candidates tie exactly (then the smaller index, by a second reduction)
                        const unsigned long long tied = __ballot(bm == wmi);
                        int wl = __builtin_ctzll(tied);
                        if (__builtin_expect((tied & (tied - 1ull)) != 0ull, 0)) {
                            const int wi = wave_min_i32_dpp(bm == wmi ? bi : 0x7fffffff);
                            wl = __builtin_ctzll(__ballot(bm == wmi && bi == wi));
                        }
                        const int win = __builtin_amdgcn_readlane(bi, wl);
                        const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), wl)),
                                    oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), wl)),
                                    oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), wl));
                        if (lane == m) { sx = ox; sy = oy; sz = oz; si = win; }
                        const fps_f2 c_x = fps_f2{ox, ox}, c_y = fps_f2{oy, oy}, c_z = fps_f2{oz, oz};
#pragma unroll
                        for (int c = 0; c < C / 2; ++c) {
                            const fps_f2 dx = qx[c] - c_x, dy = qy[c] - c_y, dz = qz[c] - c_z;
                            const fps_f2 d2 = (dx * dx + dy * dy) + dz * dz;
                            // (minimum of the bit patterns: both are >= 0 or the padding value -1, which stays)
                            qm[c].x = __int_as_float(min(__float_as_int(qm[c].x), __float_as_int(d2.x)));
                            qm[c].y = __int_as_float(min(__float_as_int(qm[c].y), __float_as_int(d2.y)));
                        }
                    }
                    if (SPLIT && wave == 0) {
                        if (lane < m) s_batch[lane] = float4{sx, sy, sz, __int_as_float(si)};
                        if (lane == 0) s_batch_m = m | (dry ? 256 : 0);
                    }
                }
                if constexpr (SPLIT) {
                    __syncthreads();
                    if (!draws) {
                        const int mm = __builtin_amdgcn_readfirstlane(s_batch_m);
                        m = mm & 255; dry = (mm & 256) != 0;
                        const float4 v = s_batch[lane];
                        sx = v.x; sy = v.y; sz = v.z; si = __float_as_int(v.w);
                    }
                }
                if (dry && m < 48) frac = fmaxf(0.3f, frac * 0.95f);
                if (m == 0) single_until = min(n_samples, s + 4);          // (cannot happen: `cur` is a candidate; never spin)
                if (lane < m) s_out[wave][(s + lane) & (BLOCK - 1)] = si;
                if (wave == 0 && lane < m) idx_out[s + lane] = si;
                flushed = s + m;
                n_adv = m;
                // the batch against the buckets: slot by slot, lane i tests sample i against the slot's box
#pragma unroll
                for (int J = 0; J < NP; ++J) {
                    auto at = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), J)); };
                    const float l0 = at(blo[0]), l1 = at(blo[1]), l2 = at(blo[2]), h0 = at(bhi[0]), h1 = at(bhi[1]), h2 = at(bhi[2]), bmJ = at(bmax);
                    const float ddx = fmaxf(fmaxf(l0 - sx, sx - h0), 0.0f), ddy = fmaxf(fmaxf(l1 - sy, sy - h1), 0.0f), ddz = fmaxf(fmaxf(l2 - sz, sz - h2), 0.0f);
                    const float lb2 = ((ddx * ddx + ddy * ddy) + ddz * ddz) * (1.0f - 4e-6f);
                    unsigned long long hit = __ballot(lane < m && !(lb2 >= bmJ));
                    if (hit) {
                        fps_f2 mm = md[J];
                        do {
                            const int i = __builtin_ctzll(hit);
                            hit &= hit - 1ull;
                            const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), i)),
                                        oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), i)),
                                        oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), i));
                            const fps_f2 c_x = fps_f2{ox, ox}, c_y = fps_f2{oy, oy}, c_z = fps_f2{oz, oz};
                            const fps_f2 dx = px[J] - c_x, dy = py[J] - c_y, dz = pz[J] - c_z;
                            const fps_f2 d2 = (dx * dx + dy * dy) + dz * dz;
                            mm.x = __int_as_float(min(__float_as_int(mm.x), __float_as_int(d2.x)));
                            mm.y = __int_as_float(min(__float_as_int(mm.y), __float_as_int(d2.y)));
                        } while (hit);
                        md[J] = mm;
                        const int b0 = __float_as_int(mm.x), b1 = __float_as_int(mm.y);
                        unsigned pj = pidx[J];
                        asm volatile("" : "+v"(pj));
                        const int i0 = (int)(pj & 0xffffu), i1 = (int)(pj >> 16);
                        const bool sel1 = (b1 > b0) | ((b1 == b0) & (i1 < i0));
                        const int lb = sel1 ? b1 : b0, li = sel1 ? i1 : i0;
                        const float4 lc = float4{sel1 ? px[J].y : px[J].x, sel1 ? py[J].y : py[J].x, sel1 ? pz[J].y : pz[J].x, 0.0f};
                        const int wmi = wave_max_i32(lb);
                        const unsigned long long tied = __ballot(lb == wmi);
                        int win;
                        if (__builtin_expect((tied & (tied - 1ull)) != 0ull, 0)) win = wave_min_i32_dpp(lb == wmi ? li : 0x7fffffff);
                        else win = __builtin_amdgcn_readlane(li, __builtin_ctzll(tied));
                        if (li == win && lb == wmi && wmi >= 0) s_cand[wave][J] = lc;
                        if (lane == J) { bmax = __int_as_float(wmi); bidx = win; }
                    }
                }
            }
        }
        if (single) {
        if (lane == 0) s_out[wave][s & (BLOCK - 1)] = cur;
        // 1. which buckets of this wave can change
        const float ddx = fmaxf(fmaxf(blo[0] - cx, cx - bhi[0]), 0.0f), ddy = fmaxf(fmaxf(blo[1] - cy, cy - bhi[1]), 0.0f),
                    ddz = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.0f);
        const float lb2 = ((ddx * ddx + ddy * ddy) + ddz * ddz) * (1.0f - 4e-6f);      // below every distance the update would compute
        unsigned mask = (unsigned)__ballot(lane < NP && !(lb2 >= bmax));
        // 2. active buckets
        while (mask) {
            const int ja = __builtin_ctz(mask);
            mask &= mask - 1u;
            fps_dispatch<NP>(ja, [&]<int J>() {
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
                unsigned pj = pidx[J];
                asm volatile("" : "+v"(pj));          // (else the 2 x NP unpacked indices are hoisted out of the sample loop into registers)
                const int i0 = (int)(pj & 0xffffu), i1 = (int)(pj >> 16);
                // the lane's better point (larger minimum, then smaller original index) and its coordinates: independent of the reduction
                const bool sel1 = b1 > b0 || (b1 == b0 && i1 < i0);
                const int lb = sel1 ? b1 : b0, li = sel1 ? i1 : i0;
                const float4 lc = float4{sel1 ? px[J].y : px[J].x, sel1 ? py[J].y : py[J].x, sel1 ? pz[J].y : pz[J].x, 0.0f};
                const int wmi = wave_max_i32(lb);                                         // wave-uniform
                const float wm = __int_as_float(wmi);
                // (a taken branch costs a lone wave about as much as fifteen instructions: ties are resolved by a second reduction, always)
                const int win = wave_min_i32_dpp(lb == wmi ? li : 0x7fffffff);
                if (li == win && lb == wmi && wmi >= 0) s_cand[wave][J] = lc;
                if (lane == J) { bmax = wm; bidx = win; }
            });
        }
        }
        // 3. best bucket of the wave: largest minimum, ties to the smaller original index (recomputed every sample: cheaper than branching
        //    around it, and a wave without work is not the one the others wait for)
        {
            const int kb = lane < NP ? __float_as_int(bmax) : -1;
            const int best = wave_max_i32(kb);
            const int bi = wave_min_i32_dpp(kb == best ? bidx : 0x7fffffff);
            const int jb = __builtin_ctzll(__ballot(kb == best && bidx == bi));
            wmd = best; widx = bi;
            const float4 c = s_cand[wave][jb < NP ? jb : 0];
            wx = c.x; wy = c.y; wz = c.z;
        }
        const int buf = (BATCH ? it++ : s) & 1;
        if (lane == 0) { s_exch[buf][wave] = float4{__int_as_float(wmd), wx, wy, wz}; s_exch_idx[buf][wave] = widx; }
        __syncthreads();
        if (__builtin_expect(single && ((s & (BLOCK - 1)) == BLOCK - 1 || s == n_samples - 1), 0)) {
            const int base = s & ~(BLOCK - 1);
            if (base + tid <= s) idx_out[base + tid] = s_out[0][tid];
            flushed = s + 1;
        }
        // all four slots are read before anything is compared (left to itself hipcc reads them one by one behind branches: four LDS
        // round trips in a row), then a two-level knock-out without branches
        static_assert(NW == 4 || NW == 8, "the knock-out below is written for four or eight waves");
        float4 e[NW]; int ei[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) { e[w] = s_exch[buf][w]; ei[w] = s_exch_idx[buf][w]; }
        asm volatile("" : "+v"(e[0].x), "+v"(e[1].x), "+v"(e[2].x), "+v"(e[3].x), "+v"(ei[0]), "+v"(ei[1]), "+v"(ei[2]), "+v"(ei[3]));
        if constexpr (NW == 8) asm volatile("" : "+v"(e[4].x), "+v"(e[5].x), "+v"(e[6].x), "+v"(e[7].x), "+v"(ei[4]), "+v"(ei[5]), "+v"(ei[6]), "+v"(ei[7]));
        const FpsSlot won = fps_knock_out<0, NW, NW>(e, ei);
        cur = won.i;
        curm = __float_as_int(won.e.x);
        cx = won.e.y; cy = won.e.z; cz = won.e.w;
        s += n_adv;
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
