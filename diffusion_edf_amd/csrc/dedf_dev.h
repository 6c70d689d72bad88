// Device-side building blocks (gfx950).  Kernel bodies are __host__ __device__ templates only so that the host pass of
// hipcc parses them; the host variants of the wave-level primitives below are inert and never called.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <utility>
#include "dedf_layout.h"

#define DEDF_DEV __host__ __device__ __forceinline__
#include "dedf_tables.h"

namespace dedf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

DEDF_DEV int lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(threadIdx.x & 63);
#else
    return 0;
#endif
}

// D(32x32) += A(32x2) * B(2x32), exact f32 (v_mfma_f32_32x32x2_f32, 64 cycles per SIMD)
DEDF_DEV f32x16 mfma32(float a, float b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#else
    (void)a; (void)b; return c;
#endif
}

// D(16x16) += A(16x4) * B(4x16), exact f32 (v_mfma_f32_16x16x4_f32, 32 cycles per SIMD).  A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], D: col = l&15, row = 4*(l>>4) + reg.
DEDF_DEV f32x4 mfma16(float a, float b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#else
    (void)a; (void)b; return c;
#endif
}
// 16-lane-row exchanges (rows = lanes 0-15 | 16-31 | 32-47 | 48-63):
//   swap16:  x' = [x0, y0, x2, y2]   y' = [x1, y1, x3, y3]          swap32:  x' = [x0, x1, y0, y1]   y' = [x2, x3, y2, y3]
// Inline asm on purpose: with ROCm 7.2 the two-result builtins __builtin_amdgcn_permlane{16,32}_swap were observed to feed
// the FIRST result to both consumers in this usage (tests/probe/mfma16_swap_probe.hip); the s_nop's are the wait states
// between a VALU write of an operand and the swap / its consumers that hipcc does not insert inside asm.
DEDF_DEV void swap16(float& x, float& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
#else
    (void)x; (void)y;
#endif
}
// five independent swap16's (the 2l+1 = 5 components of one register pair) behind ONE pair of hazard nops
DEDF_DEV void swap16x5(float (&x)[5], float (&y)[5]) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %5\n\tv_permlane16_swap_b32 %1, %6\n\tv_permlane16_swap_b32 %2, %7\n\t"
                 "v_permlane16_swap_b32 %3, %8\n\tv_permlane16_swap_b32 %4, %9\n\ts_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]));
#else
    (void)x; (void)y;
#endif
}
DEDF_DEV void swap32(float& x, float& y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
#else
    (void)x; (void)y;
#endif
}

// exchange with the lane holding the other half of this item's channel rows
DEDF_DEV float xor32(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __shfl_xor(v, 32, 64);
#else
    return v;
#endif
}

template <class F, int... I>
DEDF_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f.template operator()<I>(), ...);
}
template <int N, class F>
DEDF_DEV void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// scheduling-region fence: keeps hipcc from hoisting hundreds of weight loads to the top of a fully unrolled phase
#ifndef DEDF_FENCE
#define DEDF_FENCE 1
#endif
DEDF_DEV void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__) && DEDF_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// ---- buffer-descriptor loads: base in 4 SGPRs, ONE per-lane byte offset VGPR, everything else scalar ----------------
// (a fully unrolled kernel otherwise materialises one 64-bit VGPR address per weight load and spills hundreds of them)
struct Buf {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r;
#else
    const char* p;
#endif
};
DEDF_DEV Buf make_buf(const void* p, uint32_t bytes) {
    Buf b;
#if defined(__HIP_DEVICE_COMPILE__)
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
#else
    (void)bytes;
    b.p = static_cast<const char*>(p);
#endif
    return b;
}
DEDF_DEV f32x4 bld4(const Buf& b, int voff_bytes, int soff_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff_bytes, soff_bytes, 0));
#else
    return *reinterpret_cast<const f32x4*>(b.p + voff_bytes + soff_bytes);
#endif
}


// Copy of N floats (compile-time count) from global memory into a wave's LDS in two phases -- every request first, then every store --, so that
// a prologue costs ONE round trip to memory.  (`for (i = lane; i < n; i += 64) dst[i] = src[i]` has a lane-dependent trip count: hipcc keeps it
// a loop of load -> wait -> store round trips, ~22 of them in k_edge's prologue: several microseconds of a launch whose waves run ONE tile,
// the deployment regime of 10-20 poses.)
template <int N> struct RowRegs { float v[(N + 63) / 64]; };
template <int N> DEDF_DEV RowRegs<N> rows_request(const float* src, int lane) {
    RowRegs<N> r;
    static_for<(N + 63) / 64>([&]<int k>() { const int i = lane + 64 * k; r.v[k] = src[(N % 64 == 0 || i < N) ? i : N - 1]; });
    return r;
}
template <int N> DEDF_DEV void rows_store(float* dst, const RowRegs<N>& r, int lane) {
    static_for<(N + 63) / 64>([&]<int k>() { const int i = lane + 64 * k; if (N % 64 == 0 || i < N) dst[i] = r.v[k]; });
}
struct Wave {            // per-lane constants of the transposed-GEMM layout
    int lane, col, hi;
    int lane16;          // byte offset of this lane inside a packed-A group (64 lanes x 16 B)
    int hi64;            // byte offset of this lane's half inside a row-packed tile (2 x 16 floats)
    int lane16_r16;      // lane16 for lanes that hold rows 0-15 of a tile, out of range for the others: operands whose rows 16-31
                         // are padding are then fetched by half the lanes only (an out-of-range buffer load returns 0 without a fetch)
    int lane16_r16up;    // the same 16 rows placed as rows 16-31 of the tile: lanes of rows 16-31 fetch what the lane 16 below them would
    Buf w;               // all packed weights / row vectors of the launch
};
DEDF_DEV Wave make_wave(const void* wbuf, uint32_t wbytes) {
    Wave wv;
    wv.lane = lane_id(); wv.col = wv.lane & 31; wv.hi = wv.lane >> 5;
    wv.lane16 = wv.lane * 16; wv.hi64 = wv.hi * 64;
    wv.lane16_r16 = wv.col < 16 ? wv.lane16 : 0x7ffffff0;
    wv.lane16_r16up = wv.col >= 16 ? wv.lane16 - 256 : 0x7ffffff0;
    wv.w = make_buf(wbuf, wbytes);
    return wv;
}
// A operand of the fused stage's weight streams (layer 3, lin / sep_alpha, value)
DEDF_DEV f32x4 bldw(const Wave& wv, int voff_bytes, int soff_bytes) { return bld4(wv.w, voff_bytes, soff_bytes); }
// A operands of 4 consecutive K-steps (group g) of out tile To; matrix at float offset `off`, nG groups per tile
DEDF_DEV f32x4 lda(const Wave& wv, int off, int nG, int To, int g) {
    return bldw(wv, wv.lane16, (off + (To * nG + g) * 256) * 4);
}
// acc tile <- 16 per-row values stored [tile][hi][r] at float offset `off`
DEDF_DEV f32x16 ldrows(const Buf& b, int voff_hi64, int off, int tile) {
    f32x16 v;
    static_for<4>([&]<int G>() {
        const f32x4 t = bld4(b, voff_hi64, (off + tile * 32 + 4 * G) * 4);
        v[4 * G + 0] = t[0]; v[4 * G + 1] = t[1]; v[4 * G + 2] = t[2]; v[4 * G + 3] = t[3];
    });
    return v;
}
DEDF_DEV f32x16 ldrows(const Wave& wv, int off, int tile) { return ldrows(wv.w, wv.hi64, off, tile); }

// Make a wave-uniform scalar opaque to the optimiser at this point.  Used on weight-image offsets at the top of every
// tile: otherwise LICM hoists ~600 `offset + constant` values out of the persistent tile loop, spills them to VGPR lanes
// and pays `v_readlane; s_nop 4` between MFMAs for every weight load.
DEDF_DEV int opaque_s(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}

DEDF_DEV float opaque_s(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}

// ---- split-fp16 MFMA (v_mfma_f32_32x32x16_f16, 32 cycles): 22-bit operands, fp32 accumulate ---------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
struct HL { h8 hi, lo; };
DEDF_DEV f32x16 mfma_h(h8 a, h8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    (void)a; (void)b; return c;
#endif
}
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi): |x - hi - lo| <= 2^-22 |x| (or the fp16 subnormal step 6e-8)
// The hi half is made opaque before the residual is formed: hipcc otherwise re-derives fp16(x) a second way for the
// subtraction (v_fma_mix*_f16 beside v_cvt_pk_f16_f32) and the two do not always round alike, which breaks hi + lo == x
// by an fp16 ulp (seen as 1e-3-level errors that came and went with unrelated code changes).
#ifndef DEDF_SPLIT_MIX
#define DEDF_SPLIT_MIX 1
#endif
DEDF_DEV HL split8(const float (&x)[8]) {
    HL r;
#if defined(__HIP_DEVICE_COMPILE__) && DEDF_SPLIT_MIX
    // per pair of values: one v_cvt_pk_f16_f32 (hi halves), then v_fma_mixlo_f16 / v_fma_mixhi_f16 form fp16(x - hi) directly from
    // the packed hi halves (the mixed-precision FMA converts the fp16 source on the fly; x - hi is exact in fp32, one rounding):
    // 3 instructions per pair instead of 6 (v_cvt_pk, 2 v_cvt_f32_f16, 2 v_sub, v_cvt_pk), bit-identical halves
    unsigned hp[4], lp[4];
#define DEDF_SPLIT_PAIR(Q)                                                                          \
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"                                                   \
                 "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"             \
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                 \
                 : "=&v"(hp[Q]), "=&v"(lp[Q]) : "v"(x[2 * Q]), "v"(x[2 * Q + 1]))
    DEDF_SPLIT_PAIR(0); DEDF_SPLIT_PAIR(1); DEDF_SPLIT_PAIR(2); DEDF_SPLIT_PAIR(3);
#undef DEDF_SPLIT_PAIR
    r.hi = __builtin_bit_cast(h8, u32x4{hp[0], hp[1], hp[2], hp[3]});
    r.lo = __builtin_bit_cast(h8, u32x4{lp[0], lp[1], lp[2], lp[3]});
#else
    static_for<8>([&]<int J>() { r.hi[J] = (_Float16)x[J]; });
#if defined(__HIP_DEVICE_COMPILE__)
    f32x4 hb = __builtin_bit_cast(f32x4, r.hi);
    asm volatile("" : "+v"(hb));
    r.hi = __builtin_bit_cast(h8, hb);
#endif
    static_for<8>([&]<int J>() { r.lo[J] = (_Float16)(x[J] - (float)r.hi[J]); });
#endif
    return r;
}

// split8 of eight values whose elements 2, 3, 6, 7 are structural zeros (dedf_net.h::pad_reg): only the two live pairs are converted
DEDF_DEV HL split8z(const float (&x)[8]) {
#if defined(__HIP_DEVICE_COMPILE__) && DEDF_SPLIT_MIX
    unsigned hp[2], lp[2];
#define DEDF_SPLIT_PAIR(Q, A)                                                                       \
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"                                                   \
                 "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"             \
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                 \
                 : "=&v"(hp[Q]), "=&v"(lp[Q]) : "v"(x[A]), "v"(x[A + 1]))
    DEDF_SPLIT_PAIR(0, 0); DEDF_SPLIT_PAIR(1, 4);
#undef DEDF_SPLIT_PAIR
    HL r;
    r.hi = __builtin_bit_cast(h8, u32x4{hp[0], 0u, hp[1], 0u});
    r.lo = __builtin_bit_cast(h8, u32x4{lp[0], 0u, lp[1], 0u});
    return r;
#else
    const float y[8] = {x[0], x[1], 0.0f, 0.0f, x[4], x[5], 0.0f, 0.0f};
    return split8(y);
#endif
}
// Half-precision mode (HP: every GEMM is ONE fp16 product): only the hi halves exist -- four conversions instead of twelve instructions.
// (Round 5; until then the HP kernels formed the residual halves too and never used them.)
DEDF_DEV HL split8_hi(const float (&x)[8]) {
    HL r;
#if defined(__HIP_DEVICE_COMPILE__) && DEDF_SPLIT_MIX
    unsigned hp[4];
    asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\tv_cvt_pk_f16_f32 %2, %8, %9\n\tv_cvt_pk_f16_f32 %3, %10, %11"
                 : "=&v"(hp[0]), "=&v"(hp[1]), "=&v"(hp[2]), "=&v"(hp[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
    r.hi = __builtin_bit_cast(h8, u32x4{hp[0], hp[1], hp[2], hp[3]});
#else
    static_for<8>([&]<int J>() { r.hi[J] = (_Float16)x[J]; });
#endif
    r.lo = h8{};
    return r;
}
template <bool HP> DEDF_DEV HL split8x(const float (&x)[8]) { if constexpr (HP) return split8_hi(x); else return split8(x); }
template <bool HP> DEDF_DEV HL split8zx(const float (&x)[8]) {
    if constexpr (HP) { const float y[8] = {x[0], x[1], 0.0f, 0.0f, x[4], x[5], 0.0f, 0.0f}; return split8_hi(y); }
    else return split8z(x);
}
// float(half HALF of `hi_pair`) + float(half HALF of `lo_pair`): a split value back in fp32 (exact: the sum of a hi / lo pair IS the 22-bit operand)
// in ONE mixed-precision FMA -- hipcc turns (float)h + (float)l into two conversions and an add.
template <int HALF> DEDF_DEV float unsplit(unsigned hi_pair, unsigned lo_pair) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi_pair), "v"(lo_pair));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi_pair), "v"(lo_pair));
    return r;
#else
    (void)hi_pair; (void)lo_pair; return 0.0f;
#endif
}
// Four values -> one 16-byte word {hi(0,1), hi(2,3), lo(0,1), lo(2,3)}: the packed form of a parked chunk whose other four registers are
// structural zeros (the 8x3e block inside its 16-channel chunk, dedf_edge.h::park_chunk).  Same halves as split8 gives for these values.
DEDF_DEV f32x4 split4pk(const float (&x)[4]) {
#if defined(__HIP_DEVICE_COMPILE__) && DEDF_SPLIT_MIX
    unsigned hp[2], lp[2];
#define DEDF_SPLIT_PAIR(Q)                                                                          \
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"                                                   \
                 "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"             \
                 "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"                 \
                 : "=&v"(hp[Q]), "=&v"(lp[Q]) : "v"(x[2 * Q]), "v"(x[2 * Q + 1]))
    DEDF_SPLIT_PAIR(0); DEDF_SPLIT_PAIR(1);
#undef DEDF_SPLIT_PAIR
    return __builtin_bit_cast(f32x4, u32x4{hp[0], hp[1], lp[0], lp[1]});
#else
    const float y[8] = {x[0], x[1], 0.0f, 0.0f, x[2], x[3], 0.0f, 0.0f};
    const HL sp = split8(y);
    const u32x4 h = __builtin_bit_cast(u32x4, sp.hi), l = __builtin_bit_cast(u32x4, sp.lo);
    return __builtin_bit_cast(f32x4, u32x4{h[0], h[2], l[0], l[2]});
#endif
}

// A finished accumulator tile that the VALU reads next: pin it to architectural VGPRs here, so that the MFMAs write it there
// directly instead of into AGPRs followed by 16 v_accvgpr_read copies.
DEDF_DEV void to_vgpr(f32x16& t) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t));
#endif
}
// Make a per-lane value opaque to the optimiser at this point (no instruction): what is computed from it afterwards is not merged
// with what was computed from it before, i.e. nothing derived from it stays alive across this point.
DEDF_DEV void opaque_v(float& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}
// Returns v, but as an opaque function of `dep`: a request whose address goes through tie() cannot be issued before `dep`
// exists, and (volatile) keeps its place between the scheduling fences.  hipcc otherwise lets the pure MFMA / VALU work drift
// below the fences while the operand requests stay put, so that a whole layer's operands end up in flight (and spilled).
DEDF_DEV int tie(int v, float dep) {
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm volatile("v_and_or_b32 %0, %1, 0, %2" : "=v"(r) : "v"(dep), "v"(v));
    return r;
#else
    (void)dep; return v;
#endif
}
// Dense layer on split-fp16 MFMAs, NTO output tiles rotated, A images [To][chunk][lane][8 halves] (hi at off_h, lo at off_l),
// operands requested PD chunks ahead.   bsrc.operator()<chunk, j>() -> fp32 value of element j of the chunk (8 registers).
// HP (half-precision mode, the reference's `half_precision`): only the hi*hi term -- one MFMA and one operand image per product.
// The first PD chunks' operands can be requested ahead of time (dense_prefetch, e.g. before the previous layer's activation math):
// with one wave per SIMD nothing else hides the latency of the first requests of a layer.
template <int NTO, int PD> struct DenseRing { f32x4 h[PD][NTO], l[PD][NTO]; };
template <int NTO, int NCH, int PD = 2, bool HP = false>
DEDF_DEV DenseRing<NTO, PD> dense_prefetch(const Wave& wv, int off_h, int off_l) {
    DenseRing<NTO, PD> r{};
    static_for<PD>([&]<int k>() { if constexpr (k < NCH) static_for<NTO>([&]<int To>() {
        r.h[k][To] = lda(wv, off_h, NCH, To, k); if constexpr (!HP) r.l[k][To] = lda(wv, off_l, NCH, To, k); }); });
    return r;
}
template <int NTO, int NCH, int PD = 2, bool HP = false, class BsrcF>
DEDF_DEV void dense_rot_h(const Wave& wv, int off_h, int off_l, f32x16 (&acc)[NTO], BsrcF&& bsrc, DenseRing<NTO, PD> ring) {
    sched_fence();
    static_for<NCH>([&]<int c>() {
        f32x4 ch[NTO], cl[NTO];
        static_for<NTO>([&]<int To>() { ch[To] = ring.h[c % PD][To]; cl[To] = ring.l[c % PD][To]; });
        sched_fence();
        float t[8];
        static_for<8>([&]<int J>() { t[J] = bsrc.template operator()<c, J>(); });
        const HL b = split8x<HP>(t);
        if constexpr (c + PD < NCH) {
            const int lv = tie(wv.lane16, __builtin_bit_cast(f32x4, b.hi)[0]);
            static_for<NTO>([&]<int To>() {
                ring.h[c % PD][To] = bldw(wv, lv, (off_h + (To * NCH + c + PD) * 256) * 4);
                if constexpr (!HP) ring.l[c % PD][To] = bldw(wv, lv, (off_l + (To * NCH + c + PD) * 256) * 4);
            });
        }
        static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, ch[To]), b.hi, acc[To]); });
        if constexpr (!HP) {
            static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, ch[To]), b.lo, acc[To]); });
            static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, cl[To]), b.hi, acc[To]); });
        }
        sched_fence();
    });
}
template <int NTO, int NCH, int PD = 2, bool HP = false, class BsrcF>
DEDF_DEV void dense_rot_h(const Wave& wv, int off_h, int off_l, f32x16 (&acc)[NTO], BsrcF&& bsrc) {
    sched_fence();
    dense_rot_h<NTO, NCH, PD, HP>(wv, off_h, off_l, acc, static_cast<BsrcF&&>(bsrc), dense_prefetch<NTO, NCH, PD, HP>(wv, off_h, off_l));
}

DEDF_DEV HL split8(const float (&x)[8], float scale) {
    float t[8];
    static_for<8>([&]<int J>() { t[J] = x[J] * scale; });
    return split8(t);
}
template <bool HP> DEDF_DEV HL split8sx(const float (&x)[8], float scale) {
    float t[8];
    static_for<8>([&]<int J>() { t[J] = x[J] * scale; });
    return split8x<HP>(t);
}
// Same with the B chunks already split:  bh.operator()<chunk>() -> HL   (ring: the first PD chunks' operands, requested ahead -- dense_prefetch)
template <int NTO, int NCH, int PD = 2, bool HP = false, class BF>
DEDF_DEV void dense_rot_hp(const Wave& wv, int off_h, int off_l, f32x16 (&acc)[NTO], BF&& bh, const DenseRing<NTO, PD>& ring);
template <int NTO, int NCH, int PD = 2, bool HP = false, class BF>
DEDF_DEV void dense_rot_hp(const Wave& wv, int off_h, int off_l, f32x16 (&acc)[NTO], BF&& bh) {
    sched_fence();
    const DenseRing<NTO, PD> ring = dense_prefetch<NTO, NCH, PD, HP>(wv, off_h, off_l);
    dense_rot_hp<NTO, NCH, PD, HP>(wv, off_h, off_l, acc, static_cast<BF&&>(bh), ring);
}
template <int NTO, int NCH, int PD, bool HP, class BF>
DEDF_DEV void dense_rot_hp(const Wave& wv, int off_h, int off_l, f32x16 (&acc)[NTO], BF&& bh, const DenseRing<NTO, PD>& ring) {
    f32x4 rh[PD][NTO], rl[PD][NTO] = {};
    static_for<PD>([&]<int k>() { static_for<NTO>([&]<int To>() { rh[k][To] = ring.h[k][To]; rl[k][To] = ring.l[k][To]; }); });
    sched_fence();
    static_for<NCH>([&]<int c>() {
        f32x4 ch[NTO], cl[NTO];
        static_for<NTO>([&]<int To>() { ch[To] = rh[c % PD][To]; cl[To] = rl[c % PD][To]; });
        sched_fence();
        const HL b = bh.template operator()<c>();
        if constexpr (c + PD < NCH) {
            const int lv = tie(wv.lane16, __builtin_bit_cast(f32x4, b.hi)[0]);
            static_for<NTO>([&]<int To>() {
                rh[c % PD][To] = bldw(wv, lv, (off_h + (To * NCH + c + PD) * 256) * 4);
                if constexpr (!HP) rl[c % PD][To] = bldw(wv, lv, (off_l + (To * NCH + c + PD) * 256) * 4);
            });
        }
        static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, ch[To]), b.hi, acc[To]); });
        if constexpr (!HP) {
            static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, ch[To]), b.lo, acc[To]); });
            static_for<NTO>([&]<int To>() { acc[To] = mfma_h(__builtin_bit_cast(h8, cl[To]), b.hi, acc[To]); });
        }
        sched_fence();
    });
}
// One output tile (To, of a matrix with nCH chunks per tile) applied to NM right-hand sides that share the A operands
// (e.g. the 2l+1 components of an l-block):   bh.operator()<m, chunk>() -> HL
// (the first PD chunks' operands can be requested ahead of time -- shared_prefetch, e.g. before the VALU work that follows the PREVIOUS product: a lone
//  wave has nothing else to hide the first requests of a product behind, and the node kernel runs ~45 of them per tile)
template <int PD> struct SharedRing { f32x4 h[PD], l[PD]; };
template <int NCH, int PD = 2, bool HP = false>
DEDF_DEV SharedRing<PD> shared_prefetch(const Wave& wv, int off_h, int off_l, int nCH, int To) {
    SharedRing<PD> r{};
    static_for<PD>([&]<int k>() { if constexpr (k < NCH) {
        r.h[k] = bldw(wv, wv.lane16, (off_h + (To * nCH + k) * 256) * 4); if constexpr (!HP) r.l[k] = bldw(wv, wv.lane16, (off_l + (To * nCH + k) * 256) * 4); } });
    return r;
}
template <int NM, int NCH, int PD = 2, bool HP = false, class BF>
DEDF_DEV void dense_shared_hp(const Wave& wv, int off_h, int off_l, int nCH, int To, f32x16 (&acc)[NM], BF&& bh, const SharedRing<PD>& ring);
template <int NM, int NCH, int PD = 2, bool HP = false, class BF>
DEDF_DEV void dense_shared_hp(const Wave& wv, int off_h, int off_l, int nCH, int To, f32x16 (&acc)[NM], BF&& bh) {
    sched_fence();
    const SharedRing<PD> ring = shared_prefetch<NCH, PD, HP>(wv, off_h, off_l, nCH, To);
    dense_shared_hp<NM, NCH, PD, HP>(wv, off_h, off_l, nCH, To, acc, static_cast<BF&&>(bh), ring);
}
template <int NM, int NCH, int PD, bool HP, class BF>
DEDF_DEV void dense_shared_hp(const Wave& wv, int off_h, int off_l, int nCH, int To, f32x16 (&acc)[NM], BF&& bh, const SharedRing<PD>& ring) {
    f32x4 rh[PD], rl[PD] = {};
    static_for<PD>([&]<int k>() { rh[k] = ring.h[k]; rl[k] = ring.l[k]; });
    sched_fence();
    static_for<NCH>([&]<int c>() {
        const h8 ch = __builtin_bit_cast(h8, rh[c % PD]), cl = __builtin_bit_cast(h8, rl[c % PD]);
        sched_fence();
        HL b[NM];
        static_for<NM>([&]<int m>() { b[m] = bh.template operator()<m, c>(); });
        if constexpr (c + PD < NCH) {
            const int lv = tie(wv.lane16, __builtin_bit_cast(f32x4, b[0].hi)[0]);
            rh[c % PD] = bldw(wv, lv, (off_h + (To * nCH + c + PD) * 256) * 4);
            if constexpr (!HP) rl[c % PD] = bldw(wv, lv, (off_l + (To * nCH + c + PD) * 256) * 4);
        }
        static_for<NM>([&]<int m>() { acc[m] = mfma_h(ch, b[m].hi, acc[m]); });
        if constexpr (!HP) {
            static_for<NM>([&]<int m>() { acc[m] = mfma_h(ch, b[m].lo, acc[m]); });
            static_for<NM>([&]<int m>() { acc[m] = mfma_h(cl, b[m].hi, acc[m]); });
        }
        sched_fence();
    });
}

// Dense layer  out[NTO tiles] += W * act  with the NTO accumulators rotated inside every K-group (consecutive MFMAs hit
// different accumulators: an instruction between two MFMAs on the SAME accumulator costs a ~43-cycle bubble on gfx950).
//   bop.operator()<kg, j>() -> B operand of K-step 4*kg + j;   A operands prefetched one K-group (NTO float4) ahead.
template <int NTO, int NKG, int PD = 2, class BopF>
DEDF_DEV void dense_rot(const Wave& wv, int off, f32x16 (&acc)[NTO], BopF&& bop) {
    f32x4 ring[PD][NTO];
    static_for<PD>([&]<int k>() { if constexpr (k < NKG) static_for<NTO>([&]<int To>() { ring[k][To] = lda(wv, off, NKG, To, k); }); });
    static_for<NKG>([&]<int kg>() {
        f32x4 cur[NTO];
        static_for<NTO>([&]<int To>() { cur[To] = ring[kg % PD][To]; });
        if constexpr (kg + PD < NKG) static_for<NTO>([&]<int To>() { ring[kg % PD][To] = lda(wv, off, NKG, To, kg + PD); });
        sched_fence();
        static_for<4>([&]<int j>() {
            const float b = bop.template operator()<kg, j>();
            static_for<NTO>([&]<int To>() { acc[To] = mfma32(cur[To][j], b, acc[To]); });
        });
        sched_fence();
    });
}

// One output tile (To) applied to NM independent right-hand sides that share the A operands (e.g. the 2l+1 components of
// an l-block): NM accumulators rotate, A operands prefetched PD groups ahead.   bop<m, kg, j>() -> B operand.
template <int NM, int NKG, int PD = 2, class BopF>
DEDF_DEV void dense_shared(const Wave& wv, int off, int nG, int To, int g0, f32x16 (&acc)[NM], BopF&& bop) {
    f32x4 ring[PD];
    static_for<PD>([&]<int k>() { if constexpr (k < NKG) ring[k] = lda(wv, off, nG, To, g0 + k); });
    static_for<NKG>([&]<int kg>() {
        const f32x4 cur = ring[kg % PD];
        if constexpr (kg + PD < NKG) ring[kg % PD] = lda(wv, off, nG, To, g0 + kg + PD);
        sched_fence();
        static_for<4>([&]<int j>() {
            static_for<NM>([&]<int m>() { acc[m] = mfma32(cur[j], bop.template operator()<m, kg, j>(), acc[m]); });
        });
        sched_fence();
    });
}

// Software-pipelined stream of packed-A groups: the operand of item I+PD is requested before item I is consumed, and
// scheduling fences keep hipcc from sinking the request back next to its use (it otherwise emits load -> vmcnt(0) -> MFMA
// for every group and exposes the full L2 latency ~180 times per tile at one wave per SIMD).
//   soff.operator()<I>() -> byte offset (wave-uniform) of item I;   body.operator()<I>(f32x4 a)
template <int N, int PD, class SoffF, class BodyF>
DEDF_DEV void a_stream(const Wave& wv, SoffF&& soff, BodyF&& body) {
    f32x4 ring[PD];
    static_for<PD>([&]<int I>() { if constexpr (I < N) ring[I] = bldw(wv, wv.lane16, soff.template operator()<I>()); });
    static_for<N>([&]<int I>() {
        const f32x4 a = ring[I % PD];
        if constexpr (I + PD < N) ring[I % PD] = bldw(wv, wv.lane16, soff.template operator()<I + PD>());
        sched_fence();
        body.template operator()<I>(a);
        sched_fence();
    });
}

// One packed-A group = 4 K-steps of one 32-row output tile: acc += A[:, 4 steps] * B[4 steps, :]
DEDF_DEV void mfma_group(f32x16& acc, const f32x4 a, float b0, float b1, float b2, float b3) {
    acc = mfma32(a[0], b0, acc);
    acc = mfma32(a[1], b1, acc);
    acc = mfma32(a[2], b2, acc);
    acc = mfma32(a[3], b3, acc);
}

// ---- scalar math ---------------------------------------------------------------------------------------------------
DEDF_DEV float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// exp(x) as one v_exp_f32 on x*log2(e): relative error ~(1 + |x|) * 1e-7 (the argument product is rounded once),
// 5x fewer instructions than libm expf; arguments here are activations of O(10).  Flushes to 0 / inf like exp2f.
DEDF_DEV float fexp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#else
    return exp2f(x * 1.44269504088896340736f);
#endif
}
DEDF_DEV float sigmoidf(float x) { return rcp(1.0f + fexp(-x)); }
DEDF_DEV float siluf(float x) { return x * sigmoidf(x); }
// Staged forms over N independent values: every stage is written across all values before the next one starts.  hipcc keeps the
// source order of a fully unrolled body; written value by value the five dependent instructions of a sigmoid (v_mul, v_exp, v_add,
// v_rcp, v_mul) issue back to back through one temporary and every one of them waits for its predecessor (plus the transcendental
// wait states) -- with one wave per SIMD nothing else fills those slots.
template <int N> DEDF_DEV void sigmoid_stage(const float (&x)[N], float (&s)[N]) {        // s = 1 / (1 + exp(-x))
    static_for<N>([&]<int i>() { s[i] = x[i] * -1.44269504088896340736f; });
#if defined(__HIP_DEVICE_COMPILE__)
    static_for<N>([&]<int i>() { s[i] = __builtin_amdgcn_exp2f(s[i]); });
#else
    static_for<N>([&]<int i>() { s[i] = exp2f(s[i]); });
#endif
    static_for<N>([&]<int i>() { s[i] = 1.0f + s[i]; });
    static_for<N>([&]<int i>() { s[i] = rcp(s[i]); });
}
template <int N> DEDF_DEV void silu_stage(float (&x)[N]) {                                // x <- x * sigmoid(x)
    float s[N];
    sigmoid_stage<N>(x, s);
    static_for<N>([&]<int i>() { x[i] = x[i] * s[i]; });
}
// normalize2mom-wrapped activations (reference equiformer/fast_activation.py:69)
DEDF_DEV float silu_n(float x) { return siluf(x) * kNormSilu; }
DEDF_DEV float sigmoid_n(float x) { return sigmoidf(x) * kNormSigmoid; }
// SmoothLeakyReLU(0.2) (reference fast_activation.py:14-23) x normalize2mom
DEDF_DEV float slrelu_n(float x) {
    const float x1 = 0.6f * x;
    const float x2 = 0.4f * x * (2.0f * sigmoidf(x) - 1.0f);
    return (x1 + x2) * kNormSlrelu;
}
// soft_step (reference radial_func.py:15-17, n = 3)
DEDF_DEV float soft_step(float x) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    const float x3 = x * x * x;
    return 4.0f * x3 - 3.0f * x3 * x;
}
// sin(x) (want_cos = 0) or cos(x) (want_cos = 1): 3-term Cody-Waite reduction by pi/2 with FMA + cephes
// minimax polynomials on [-pi/4, pi/4]; ~1 ulp for |x| < 1e5, branch-free (arguments here are <= ~1e3 rad).
DEDF_DEV float sin_or_cos(float x, int want_cos) {
    const float fn = rintf(x * 0.636619772367581343f);
    const int n = (int)fn + want_cos;
    float r = fmaf(fn, -1.5707963705062866f, x);
    r = fmaf(fn, 4.371138828673793e-08f, r);
    r = fmaf(fn, 1.7151245100058819e-15f, r);
    const float r2 = r * r;
    const float s = fmaf(r * r2, fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r);
    const float c = fmaf(r2 * r2, fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                         fmaf(-0.5f, r2, 1.0f));
    const float v = (n & 1) ? c : s;
    return (n & 2) ? -v : v;
}

DEDF_DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
DEDF_DEV void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

}  // namespace dedf
