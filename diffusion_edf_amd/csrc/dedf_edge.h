// Fused per-edge pipeline of the equivariant graph attention (one wave = one tile of 32 edges of one scale).
//
// Restates, per edge, reference graph_parser.py:146-224 (geometry, soft cut-offs, length embedding, spherical
// harmonics), multiscale_tensor_field.py:225-234 (edge pre-linear with the time embedding), and
// graph_attention.py:231-247 (radial MLP -> depth-wise TP -> {sep_alpha | lin -> Gate -> depth-wise TP -> lin}).
// Nothing per-edge reaches HBM: the 480 radial weights, the two 1568-float TP outputs and all activations stay in registers /
// LDS, and the results leave the kernel as ONE record per run of same-destination edges of a tile (softmax-weighted mean value
// in internal layout + log-sum-exp of the logits per head, 244 floats at lmax 2), see "joint-softmax partials" below.
//
// Layout: lane = (edge column = lane & 31, row half hi = lane >> 5); see dedf_layout.h.  Every dense layer is
// D[out][edge] += W[out][k] * act[k][edge] as three v_mfma_f32_32x32x16_f16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate)
// with the previous layer's accumulator registers split into fp16 hi / lo halves as B operands; weights stream from L2 as
// pre-split, pre-permuted 16-byte A operands per lane through one buffer descriptor.
#pragma once
#include "dedf_dev.h"
#include "dedf_net.h"

namespace dedf {

struct EdgeParams {
    // graph
    const float* key_x;       // [sum N_s][3]
    const float* qpos;        // [N_d][3] transformed query positions
    const int* edge_src;      // [E] global key index
    const int* edge_dst;      // [E]
    const int* tile_info;     // [0..n_scales] tile prefix, [16..16+n_scales] edge prefix
    const float* msg;         // [sum N_s][D]  source message (LN + LinearRS of key features), reference layout
    uint32_t msg_bytes;
    const float* msg_dst;     // UNet layer: [N_d][D] destination message (linear_dst), reference layout.  Score head with query_time_encoding (QT): the
    uint32_t msg_dst_bytes;   // time rows of dedf_misc.h::k_time_query -- [rows][kQueryTimeRow], the first 64 floats of a row join the 0e block of the message
    int qd_pose_stride;       // QT: floats between the rows of consecutive poses; 0 when every pose shares the time (sampler)
    float ln_inv_n[2], ln_pad[2];   // UNet layer only: 1 / true width and number of padded channels of the radial MLP's two LayerNorms
    const float* tb;          // [(nT|1)][n_scales][F0] row-packed: W_pre[:,64:] c_t + b_pre  (b_pre alone when F0 = 64)
    uint32_t tb_bytes;
    int tb_pose_stride;       // floats; 0 when every pose shares the time (sampler)
    int nQ, n_scales;
    // per-scale length encoders / cut-offs
    float radius[kMaxScales];      // <= 0 : infinite scale
    float cut_begin[kMaxScales];   // 0.8 r
    float cut_div[kMaxScales];     // r - 0.8 r
    float ns_lo, ns_div;           // non-scalar SH cut-off: soft_step((d - ns_lo) / ns_div)
    float len_enc_max_r;
    // packed weights: one buffer, float offsets
    const float* W;
    uint32_t W_bytes;
    int o_enc;                // [n_scales][3][2][32]  mean | 1/std | weight in (hi, s) order; infinite: freq[32] first
    int o_A_pre, o_A_pre_l;   // split-fp16 images [n_scales][F0/32 tiles][4 chunks][64][8 halves] (hi | lo)
    int o_A_r1, o_A_r1_l, o_b_r1, o_g_r1, o_be_r1;
    int o_A_r2, o_A_r2_l, o_b_r2, o_g_r2, o_be_r2;
    int o_A_r3, o_A_r3_l, o_off_r3;
    int o_S_lin;              // split-fp16 A-operand stream (dedf_pack.h::pack_dtp_stream): l3 = 0 -> lin0 rows + alpha rows; l3 >= 1: mul(l3) rows
    int o_b_r0;               // row-packed bias over the l3 = 0 row space
    int o_S_val;              // same for sep_value.lin with the shared DTP weights folded in
    float w_unscale, u_scale, c_lin[4], c_val[4];     // power-of-two operand scaling of the split-fp16 GEMMs (dedf_pack.h::EdgeOffsets)
    int o_b_val0;             // row-packed (64)
    int o_alpha_dot;          // row-packed over the two alpha tiles
    const float* key_w;       // optional [sum N_s]: key-point attention weights (use_src_point_attn: alpha *= w_src after the softmax)
    float* out;               // [E][edge_rec]: ONE record per (destination, tile) segment, stored at the segment's first edge:
                              //   value = softmax-weighted mean of the segment's edge values, logit = log-sum-exp of its logits
    float* dbg_out;           // optional [E][edge_rec] per-edge records (value, logits) for the stage tests
    float* dbg_w;             // optional [E][WN] dump of the radial weights (tests)
    unsigned long long* phase_prof;   // optional [grid][16] per-wave phase cycle sums (built with -DDEDF_PHASE_PROF)
    // Radial table (edge_tile MODE 1 / 2): when every pose of a launch shares the diffusion time -- the sampler -- everything in front of
    // layer 3 of the radial network (length encoding, pre-linear, layers 1 and 2 with their LayerNorm + SiLU) is a function of (scale,
    // edge length) alone.  k_radial_table evaluates it with the kernel's own code on a fine length grid once per launch
    // ([row][half][32] floats: row g of scale n is the node at length (g - 1) * rtab_step[n]; rtab_n[n] + 3 rows from rtab_row0[n]); the
    // edge kernel reads four neighbouring rows per edge and interpolates (4-point Lagrange: error ~ 0.023 step^4 |d4f/dlen4|).
    const float* rtab;
    uint32_t rtab_bytes;
    float* rtab_out;
    int rtab_row0[kMaxScales], rtab_n[kMaxScales];
    float rtab_step[kMaxScales], rtab_inv_step[kMaxScales];
    // Accuracy guard of the table (edge_tile MODE 3, k_radial_check): the front is ALSO evaluated exactly at the midpoint of every grid
    // interval and compared with what the 4-point interpolation of the table gives there; the largest |interpolated - exact| activation of
    // a scale accumulates (atomic max on the float's bits) in rtab_err[scale].  The table-reading kernel takes a scale's front from the table
    // only while rtab_err[scale] <= rtab_err_bound[scale], per edge otherwise (visible in dedf_stats.rtab_fallback).
    unsigned* rtab_err;
    float rtab_err_bound[kMaxScales];
    // Launch gate (dedf_score with one time for all poses): when `gate` is set the kernel runs only if (*gate != 0) == (gate_want != 0).  The word is
    // tile_info[kFlagTimeVaries], written by k_time_bias: dedf_score enqueues the table path (gate_want 0) AND the per-pose-time path (gate_want 1)
    // without reading the times back; the one that does not apply returns at once.
    const int* gate;
    int gate_want;
};
DEDF_DEV bool edge_gate_closed(const EdgeParams& P) { return P.gate != nullptr && (*P.gate != 0) != (P.gate_want != 0); }

template <int L> struct SH {           // spherical harmonics of one edge, non-scalar blocks already cut off
    float y0[1], y1[3], y2[5], y3[7];
    template <int l> DEDF_DEV const float* get() const {
        if constexpr (l == 0) return y0; else if constexpr (l == 1) return y1;
        else if constexpr (l == 2) return y2; else return y3;
    }
};

// ---- per-row vectors (biases, LayerNorm affine, layer-3 offsets, alpha_dot) live in LDS ---------------------------------
// They are the same for every tile and identical inside a half-wave; read through the vector L1 every tile they were 41
// row tiles x 4 KiB = 29 % of the kernel's L1 delivery for 5.5 KiB of data.  Each wave copies them once into its own LDS
// (k_edge prologue) and reads them back with broadcast ds_read_b128.  Offsets in floats, layout [tile][half][16] as packed.
// FRONT = false: the rows of the radial network's FRONT (layers 1-2: biases, LayerNorm affine; the length-encoder constants) stay in global
// memory; big_rows_in_lds = false: so do the layer-3 offsets (3.5 KB at lmax 3) and the lin / sep_alpha biases.  The lmax-3 kernels do both:
// with the parked operands at 39 KB per wave (dedf_net.h::park_phys) only val0 / alpha_dot (512 B) fit beside them if FOUR waves are to share
// a CU's 160 KB -- with the rows in LDS it was three, i.e. one SIMD of every CU idle.
template <int L, int MODE> constexpr bool front_rows_in_lds() { return L < 3; }
template <int L> constexpr bool big_rows_in_lds() { return L < 3; }
template <int L, bool FRONT = true> struct RowsLds {
    static constexpr bool BIG = big_rows_in_lds<L>();
    static constexpr int b1 = 0, g1 = 128, be1 = 256, b2 = 384, g2 = 448, be2 = 512, off3 = FRONT ? 576 : 0;
    static constexpr int b0 = off3 + (BIG ? rup(dtp_wn<L>(), 32) : 0), val0 = b0 + (BIG ? r0_tiles<L>() * 32 : 0), adot = val0 + 64;
    static constexpr int enc = adot + 64, total = enc + (FRONT ? 192 : 0);      // enc: length-encoder constants of the scale being processed
};
template <int L, bool FRONT = true> DEDF_DEV float* rows_lds() {
    __shared__ __attribute__((aligned(16))) float rows[RowsLds<L, FRONT>::total];
    return rows;
}
// accumulator tile <- 16 per-row values of row tile `tile` of the vector at LDS offset `off`
DEDF_DEV f32x16 ldrows_lds(const float* rows, int hi, int off, int tile) {
    f32x16 v;
    int lane_off = hi * 16;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane_off));       // keeps the reads where they are used (hipcc otherwise gathers every row read of the
                                             // tile at one point and spills their common address register)
#endif
    const f32x4* p = reinterpret_cast<const f32x4*>(rows + off + tile * 32 + lane_off);
    static_for<4>([&]<int G>() { const f32x4 t = p[G]; v[4 * G] = t[0]; v[4 * G + 1] = t[1]; v[4 * G + 2] = t[2]; v[4 * G + 3] = t[3]; });
    return v;
}

// LayerNorm over NT*32 channels of one item (rows split over lane and lane^32) followed by SiLU
// MASKED (UNet layers whose true width is smaller than the instantiated one): only the first n channels are real, the others are exactly
// 0 on entry (zero-padded weights and biases) and must stay out of the statistics: mean = sum / n, and every padded channel
// contributes exactly mean^2 to sum (x - mean)^2, which is taken out again (inv_n = 1 / n, n_pad = NT * 32 - n); their affine
// parameters are zero, so they leave as SiLU(0) = 0.
template <int NT, bool MASKED = false>
DEDF_DEV void ln_silu(f32x16 (&x)[NT], const Wave& wv, const float* r_gamma, const float* r_beta, float inv_n = 0.0f, float n_pad = 0.0f) {
    float s = 0.0f;
    static_for<NT>([&]<int T>() { static_for<16>([&]<int R>() { s += x[T][R]; }); });
    s += xor32(s);
    const float mean = MASKED ? s * inv_n : s * (1.0f / (NT * 32));
    float v = 0.0f;
    static_for<NT>([&]<int T>() { static_for<16>([&]<int R>() { const float d = x[T][R] - mean; v += d * d; }); });
    v += xor32(v);
    if constexpr (MASKED) v = fmaxf(v - n_pad * (mean * mean), 0.0f);
    const float rstd = 1.0f / sqrtf(v * (MASKED ? inv_n : 1.0f / (NT * 32)) + 1e-5f);
    static_for<NT>([&]<int T>() {
        const f32x16 g = ldrows_lds(r_gamma, wv.hi, 0, T), b = ldrows_lds(r_beta, wv.hi, 0, T);
        float y[16];
        static_for<16>([&]<int R>() { y[R] = (x[T][R] - mean) * rstd * g[R] + b[R]; });
        silu_stage<16>(y);
        static_for<16>([&]<int R>() { x[T][R] = y[R]; });
    });
}

#if defined(DEDF_PHASE_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define DEDF_STAMP(i)                                                  \
    do {                                                               \
        sched_fence();                                                 \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        pacc[i] += t_now - t_last;                                     \
        t_last = t_now;                                                \
        sched_fence();                                                 \
    } while (0)
#else
#define DEDF_STAMP(i) do { } while (0)
#endif
#if defined(DEDF_PHASE_PROF)
#define DEDF_PROF_ARG , unsigned long long (&pacc)[16]
#else
#define DEDF_PROF_ARG
#endif

// ---- linear layers fed by a depth-wise TP, on split-fp16 MFMAs ---------------------------------------------------------
// Work unit = chunk of 16 DTP channels (dedf_net.h::dtp_item): the lane-local Clebsch-Gordan products of a chunk are split
// into hi / lo halves (B operands), the weights come as a pre-split A-operand stream, every product is hi*hi + hi*lo + lo*hi.
template <int L> struct BOpsH { h8 hi[2 * L + 1], lo[2 * L + 1]; };
struct AItem { f32x4 h[2], l[2]; };
template <int L, int NT0, int I, bool HP = false, bool S = false>
DEDF_DEV AItem load_item(const Wave& wv, int o_str) {
    AItem a{};
    constexpr DtpItem it = dtp_item<L, S>(I, NT0);
    if constexpr (it.ntile > 0) {
        static_for<it.ntile>([&]<int n>() {
            // rows 16-31 of this operand are padding: the 16-channel l = 2 outputs, and the last lin tile (112 = 3.5 x 32 rows)
            constexpr int l3 = dtp_pos_l3<L, S>(it.pos);
            constexpr bool half_rows = l3 >= 2 || (l3 == 0 && NT0 == r0_tiles<L>() && lin0_rows<L>() % 32 == 16 && 2 * it.t + n == lin0_rows<L>() / 32);
            const int lv = half_rows ? wv.lane16_r16 : wv.lane16;
            a.h[n] = bldw(wv, lv, (o_str + (it.slot + n) * 512) * 4);
            if constexpr (!HP) a.l[n] = bldw(wv, lv, (o_str + (it.slot + n) * 512 + 256) * 4);
            if constexpr (acc_paired<L>(l3)) {      // the same 16 rows once more, as rows 16-31 of the tile (odd components / second tile half)
                a.h[1] = bldw(wv, wv.lane16_r16up, (o_str + it.slot * 512) * 4);
                if constexpr (!HP) a.l[1] = bldw(wv, wv.lane16_r16up, (o_str + it.slot * 512 + 256) * 4);
            }
        });
    }
    return a;
}
// MFMAs of the chunk at walk position C: l3 = 0 -> output tiles acc0[0 .. NT0) two at a time, l3 = 1 / 2 -> acc1[m] / acc2[m] (one tile
// per component; the 16-channel l = 2 outputs only fill rows 0-15, i.e. registers 0-7).  The three terms are issued
// term-major so that consecutive MFMAs hit different accumulators.
// Output-side chunks (dedf_net.h::dtp_path_out_side) accumulate G_i = W . (w x_i) into `go`: one tile per input component i -- or, for
// scalar inputs (one component), one tile per product term, so that consecutive MFMAs still hit different accumulators.
template <int L, int NT0, int C, bool HP, int PD, bool S = false>
DEDF_DEV void mfma_chunk(const Wave& wv, int o_str, AItem (&ring)[PD], const BOpsH<L>& bo, f32x16 (&acc0)[NT0], f32x16 (&acc1)[3], f32x16 (&acc2)[5],
                         f32x16 (&acc3)[7], f32x16 (&go)[3]) {
    constexpr int l3 = dtp_pos_l3<L, S>(C), I0 = dtp_item_first<L, S>(C, NT0), NI = l3 == 0 ? cdiv(NT0, 2) : 1;
    if constexpr (S && l3 >= 1) {
        // edge-aligned frame (dedf_net.h::make_dtp_walk_so2): term t of the path is ONE product  acc[k_t] += A . B_t  with the chunk's single A slot
        // (the path's reference coefficient folded in); term-major like the general form, so that consecutive MFMAs hit different accumulators
        // (paired tiles, lmax 3 / l3 >= 2: component K lives in rows 16 (K % 2) .. of tile K / 2 and takes the A operand placed there, see below)
        constexpr PathInfo pi = dtp_pos_path<L, S>(C);
        constexpr int NTm = kSo2NT[pi.l1][pi.l2][pi.l3];
        constexpr bool PR = acc_paired<L>(l3);
        const AItem a = ring[I0 % PD];
        ring[I0 % PD] = load_item<L, NT0, I0 + PD, HP, S>(wv, o_str);
        const h8 ah = __builtin_bit_cast(h8, a.h[0]), al = __builtin_bit_cast(h8, a.l[0]), bh = __builtin_bit_cast(h8, a.h[1]), bl = __builtin_bit_cast(h8, a.l[1]);
        auto& accm = [&]() -> auto& { if constexpr (l3 == 1) return acc1; else if constexpr (l3 == 2) return acc2; else return acc3; }();
        static_for<NTm>([&]<int t>() { constexpr int K = kSo2K[pi.l1][pi.l2][pi.l3][t], T = PR ? K / 2 : K; accm[T] = mfma_h(PR && K % 2 ? bh : ah, bo.hi[t], accm[T]); });
        if constexpr (!HP) {
            static_for<NTm>([&]<int t>() { constexpr int K = kSo2K[pi.l1][pi.l2][pi.l3][t], T = PR ? K / 2 : K; accm[T] = mfma_h(PR && K % 2 ? bh : ah, bo.lo[t], accm[T]); });
            static_for<NTm>([&]<int t>() { constexpr int K = kSo2K[pi.l1][pi.l2][pi.l3][t], T = PR ? K / 2 : K; accm[T] = mfma_h(PR && K % 2 ? bl : al, bo.hi[t], accm[T]); });
        }
    } else
    if constexpr (dtp_pos_out<L, S>(C)) {
        constexpr int d1 = 2 * dtp_pos_path<L, S>(C).l1 + 1;
        static_assert(d1 == 1 || d1 == 3, "output-side paths have input degree 0 or 1");
        constexpr bool first = dtp_pos_path_first<L, S>(C);
        const AItem a = ring[I0 % PD];
        ring[I0 % PD] = load_item<L, NT0, I0 + PD, HP, S>(wv, o_str);
        const h8 ah = __builtin_bit_cast(h8, a.h[0]), al = __builtin_bit_cast(h8, a.l[0]);
        const f32x16 zero = {};
        if constexpr (acc_paired<L>(l3)) {
            // 16-row outputs: two G tiles instead of three (see the input-side branch below).  d1 = 3: components 0 | 1 share tile 0, component 2
            // has tile 1;  d1 = 1: the product terms hi*hi | hi*lo share tile 0, lo*hi has tile 1.  contract_out reads them accordingly.
            const h8 bh = __builtin_bit_cast(h8, a.h[1]), bl = __builtin_bit_cast(h8, a.l[1]);
            if constexpr (d1 == 1) {
                go[0] = mfma_h(ah, bo.hi[0], first ? zero : go[0]);
                if constexpr (!HP) { go[1] = mfma_h(al, bo.hi[0], first ? zero : go[1]); go[0] = mfma_h(bh, bo.lo[0], go[0]); }
            } else {
                go[0] = mfma_h(ah, bo.hi[0], first ? zero : go[0]);
                go[1] = mfma_h(ah, bo.hi[2], first ? zero : go[1]);
                go[0] = mfma_h(bh, bo.hi[1], go[0]);
                if constexpr (!HP) {
                    go[0] = mfma_h(ah, bo.lo[0], go[0]); go[1] = mfma_h(ah, bo.lo[2], go[1]); go[0] = mfma_h(bh, bo.lo[1], go[0]);
                    go[0] = mfma_h(al, bo.hi[0], go[0]); go[1] = mfma_h(al, bo.hi[2], go[1]); go[0] = mfma_h(bl, bo.hi[1], go[0]);
                }
            }
        } else if constexpr (d1 == 1) {
            go[0] = mfma_h(ah, bo.hi[0], first ? zero : go[0]);
            if constexpr (!HP) { go[1] = mfma_h(ah, bo.lo[0], first ? zero : go[1]); go[2] = mfma_h(al, bo.hi[0], first ? zero : go[2]); }
        } else {
            static_for<3>([&]<int i>() { go[i] = mfma_h(ah, bo.hi[i], first ? zero : go[i]); });
            if constexpr (!HP) {
                static_for<3>([&]<int i>() { go[i] = mfma_h(ah, bo.lo[i], go[i]); });
                static_for<3>([&]<int i>() { go[i] = mfma_h(al, bo.hi[i], go[i]); });
            }
        }
    } else
    static_for<NI>([&]<int t>() {
        constexpr int I = I0 + t;
        const AItem a = ring[I % PD];
        ring[I % PD] = load_item<L, NT0, I + PD, HP, S>(wv, o_str);
        if constexpr (l3 == 0) {
            constexpr int nt = dtp_item<L, S>(I, NT0).ntile;
            static_for<nt>([&]<int n>() { acc0[2 * t + n] = mfma_h(__builtin_bit_cast(h8, a.h[n]), bo.hi[0], acc0[2 * t + n]); });
            if constexpr (!HP) {
                static_for<nt>([&]<int n>() { acc0[2 * t + n] = mfma_h(__builtin_bit_cast(h8, a.h[n]), bo.lo[0], acc0[2 * t + n]); });
                static_for<nt>([&]<int n>() { acc0[2 * t + n] = mfma_h(__builtin_bit_cast(h8, a.l[n]), bo.hi[0], acc0[2 * t + n]); });
            }
        } else {
            constexpr int d3 = 2 * l3 + 1;
            auto& accm = [&]() -> auto& { if constexpr (l3 == 1) return acc1; else if constexpr (l3 == 2) return acc2; else return acc3; }();
            const h8 ah = __builtin_bit_cast(h8, a.h[0]), al = __builtin_bit_cast(h8, a.l[0]);
            if constexpr (acc_paired<L>(l3)) {
                // Two components per accumulator tile: a 16-channel output block fills rows 0-15 only, so component 2 k goes to rows 0-15 of
                // tile k (A = [W; 0]) and component 2 k + 1 to rows 16-31 (A = [0; W]: the same operand bytes requested by the other half
                // of the lanes).  Same MFMAs, (d3 + 1) / 2 instead of d3 live accumulator tiles.  Even components first: consecutive MFMAs
                // hit different tiles.
                const h8 bh = __builtin_bit_cast(h8, a.h[1]), bl = __builtin_bit_cast(h8, a.l[1]);
                static_for<d3>([&]<int K>() { if constexpr (K % 2 == 0) accm[K / 2] = mfma_h(ah, bo.hi[K], accm[K / 2]); });
                static_for<d3>([&]<int K>() { if constexpr (K % 2 == 1) accm[K / 2] = mfma_h(bh, bo.hi[K], accm[K / 2]); });
                if constexpr (!HP) {
                    static_for<d3>([&]<int K>() { if constexpr (K % 2 == 0) accm[K / 2] = mfma_h(ah, bo.lo[K], accm[K / 2]); });
                    static_for<d3>([&]<int K>() { if constexpr (K % 2 == 1) accm[K / 2] = mfma_h(bh, bo.lo[K], accm[K / 2]); });
                    static_for<d3>([&]<int K>() { if constexpr (K % 2 == 0) accm[K / 2] = mfma_h(al, bo.hi[K], accm[K / 2]); });
                    static_for<d3>([&]<int K>() { if constexpr (K % 2 == 1) accm[K / 2] = mfma_h(bl, bo.hi[K], accm[K / 2]); });
                }
            } else {
            static_for<d3>([&]<int K>() { accm[K] = mfma_h(ah, bo.hi[K], accm[K]); });
            if constexpr (!HP) {
                static_for<d3>([&]<int K>() { accm[K] = mfma_h(ah, bo.lo[K], accm[K]); });
                static_for<d3>([&]<int K>() { accm[K] = mfma_h(al, bo.hi[K], accm[K]); });
            }
            }
        }
    });
}
// split the chunk's fp32 products v[m][8] into B operands
// (PADZ: elements 2, 3, 6, 7 are the zero padding of the chunk's input channels, dedf_net.h::pad_reg)
template <int L, int l3, bool PADZ = false, bool HP = false>
DEDF_DEV void split_chunk(const float (&v)[2 * l3 + 1][8], BOpsH<L>& o) {
    static_for<2 * l3 + 1>([&]<int K>() {
        HL sp;
        if constexpr (PADZ) sp = split8zx<HP>(v[K]); else sp = split8x<HP>(v[K]);
        o.hi[K] = sp.hi; o.lo[K] = sp.lo;
    });
}

// once per wave, before its first tile: copy the row vectors into LDS (all requests, then all stores: dedf_dev.h::rows_request)
template <int L, int H1 = 128, int H2 = 64, bool FRONT = true>
DEDF_DEV void edge_rows_to_lds(const EdgeParams& P, const Wave& wv) {
    using RL = RowsLds<L, FRONT>;
    float* rows = rows_lds<L, FRONT>();
    const float* const W = P.W;
    const int lane = wv.lane;
    constexpr int NO3 = RL::BIG ? rup(dtp_wn<L>(), 32) : 64, NB0 = RL::BIG ? r0_tiles<L>() * 32 : 64;
    RowRegs<H1> f1[3];
    RowRegs<H2> f2[3];
    RowRegs<NO3> o3;
    RowRegs<NB0> b0;
    if constexpr (FRONT) {
        f1[0] = rows_request<H1>(W + P.o_b_r1, lane); f1[1] = rows_request<H1>(W + P.o_g_r1, lane); f1[2] = rows_request<H1>(W + P.o_be_r1, lane);
        f2[0] = rows_request<H2>(W + P.o_b_r2, lane); f2[1] = rows_request<H2>(W + P.o_g_r2, lane); f2[2] = rows_request<H2>(W + P.o_be_r2, lane);
    }
    if constexpr (RL::BIG) { o3 = rows_request<NO3>(W + P.o_off_r3, lane); b0 = rows_request<NB0>(W + P.o_b_r0, lane); }
    const RowRegs<64> v0 = rows_request<64>(W + P.o_b_val0, lane), ad = rows_request<64>(W + P.o_alpha_dot, lane);
    sched_fence();
    if constexpr (FRONT) {
        rows_store<H1>(rows + RL::b1, f1[0], lane); rows_store<H1>(rows + RL::g1, f1[1], lane); rows_store<H1>(rows + RL::be1, f1[2], lane);
        rows_store<H2>(rows + RL::b2, f2[0], lane); rows_store<H2>(rows + RL::g2, f2[1], lane); rows_store<H2>(rows + RL::be2, f2[2], lane);
    }
    if constexpr (RL::BIG) { rows_store<NO3>(rows + RL::off3, o3, lane); rows_store<NB0>(rows + RL::b0, b0, lane); }
    rows_store<64>(rows + RL::val0, v0, lane); rows_store<64>(rows + RL::adot, ad, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// whenever a wave moves on to tiles of another scale: that scale's length-encoder constants (192 floats)
template <int L, bool FRONT = true>
DEDF_DEV void edge_enc_to_lds(const EdgeParams& P, const Wave& wv, int scale) {
    if constexpr (FRONT) {
        float* rows = rows_lds<L, FRONT>();
        const RowRegs<192> e = rows_request<192>(P.W + P.o_enc + scale * 192, wv.lane);
        rows_store<192>(rows + RowsLds<L, FRONT>::enc, e, wv.lane);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
}

// H1, H2: hidden widths of the radial MLP (fc_neurons[1:]): 128, 64 in the panda_* and sapien pick configs, 32, 32 in sapien place_*
// UN: the edge pipeline of a UNet layer (block.EquiformerBlock + GraphAttentionMLP, block.py:141-174, graph_attention.py:84-122):
//     message = linear_src(f_src)[src] + linear_dst(f_dst)[dst], the radial MLP reads the radial basis directly (no pre-linear, no
//     time), GaussianRadialBasisLayerFiniteCutoff instead of GaussianRadialBasis, SH without the non-scalar cut-off, no edge logit
// Geometry of the NEXT tile, carried across the persistent tile loop by the table-reading kernel (MODE 1): with the radial network's front
// gone, a tile would otherwise start with three dependent round trips to memory (edge indices -> coordinates -> table rows) and nothing to
// put under them.  The indices are requested at the top of the previous tile, the coordinates in its middle.
struct GeoPre { int ok, src, dst; float vx, vy, vz; };

// MODE 0: everything per edge.  MODE 1: the radial network's front comes from the radial table (EdgeParams::rtab) for every tile whose
// lengths lie inside the table, per edge otherwise (the all-pairs scale has no a-priori bound).  MODE 3: accuracy check of the table --
// "edge" e of the tile is the MIDPOINT of grid interval e of `scale`: exact front there against the interpolated table.  MODE 2: table generator -- "edge" e of
// the tile is table row e of `scale`; the tile ends after layer 2's activation.
// NW: narrow UNet level (dedf_net.h::pad_live): the lane-local work on the structurally zero channels is skipped
// SO2: both depth-wise TPs in the edge-aligned frame (diffusion_edf_amd/so2.py; dedf_net.h::make_dtp_walk_so2 / make_sval_walk): the gathered source
//      rows are rotated so that the edge is the polar axis (Rot<l>::in: X(gamma) J X(beta) J on cos / sin of m gamma, m beta), where the SH has only
//      its m = 0 component and every path is one product per output component; lin, Gate, lin2 and sep_alpha commute with the rotation, the value
//      is rotated back (Rot<l>::out) before the segmented reduction.  Same result as the general form up to fp32 rounding; needs the image packed
//      for it (dedf_pack.h::pack_edge<L, true>).
template <int L> struct Trig { float cg[L], sg[L], cb[L], sb[L]; };
template <int L, int F0, bool HP = false, int H1 = 128, int H2 = 64, bool UN = false, int MODE = 0, bool NW = false, bool SO2 = false, bool QT = false>
DEDF_DEV void edge_tile(const EdgeParams& P, const Wave& wv, int scale, int e0, int n_valid, GeoPre& geo, int e_next DEDF_PROF_ARG) {
    static_assert((H1 == 128 && H2 == 64) || (H1 == 32 && H2 == 32), "radial MLP widths of the shipped configs");
    static_assert(!SO2 || MODE <= 1, "edge-frame form: the per-edge and the table-reading kernels");
    static_assert(!UN || F0 == 64, "UNet layer: the radial MLP reads the 64 radial-basis channels");
    static_assert(MODE == 0 || !UN, "the radial table is the sampler's");
    static_assert(!NW || UN, "NW is a UNet-layer shape");
    // activation stage on the registers of a run that hold true channels only (the others are structural zeros and stay 0)
    auto on_live = [&]<int l, int N, class St>(float (&v)[N], St&& st) {
        constexpr int NL = live_count<L, NW>(l, N);
        if constexpr (NL == N) st.template operator()<N>(v);
        else {
            float w[NL];
            static_for<N>([&]<int R>() { if constexpr (!pad_reg<L, NW>(l, R)) w[live_count<L, NW>(l, R)] = v[R]; });
            st.template operator()<NL>(w);
            static_for<N>([&]<int R>() { if constexpr (pad_reg<L, NW>(l, R)) v[R] = 0.0f; else v[R] = w[live_count<L, NW>(l, R)]; });
        }
    };
#if defined(DEDF_PHASE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    constexpr int D = feat_dim<L>();
    constexpr int REC = edge_rec<L>();
    constexpr int WN = dtp_wn<L>();
    constexpr int NWT = cdiv(WN, 32);      // (lmax 3: 880 rows = 27.5 tiles, the last half tile is zero rows)
    constexpr int NR0 = r0_tiles<L>();
    const int hi = wv.hi;
    constexpr bool FRONT = front_rows_in_lds<L, MODE>();
    using RL = RowsLds<L, FRONT>;
    const float* const rows = rows_lds<L, FRONT>();
    // the front's row vectors: this wave's LDS copy, or global memory (see RowsLds)
    const float* const r_b1 = FRONT ? rows + RL::b1 : P.W + P.o_b_r1, * const r_g1 = FRONT ? rows + RL::g1 : P.W + P.o_g_r1, * const r_be1 = FRONT ? rows + RL::be1 : P.W + P.o_be_r1;
    const float* const r_b2 = FRONT ? rows + RL::b2 : P.W + P.o_b_r2, * const r_g2 = FRONT ? rows + RL::g2 : P.W + P.o_g_r2, * const r_be2 = FRONT ? rows + RL::be2 : P.W + P.o_be_r2;
    // weight-image offsets, re-materialised per tile (see opaque_s)
    const int o_A_r1 = opaque_s(P.o_A_r1), o_A_r2 = opaque_s(P.o_A_r2), o_A_r3 = opaque_s(P.o_A_r3);
    const int o_A_r1_l = opaque_s(P.o_A_r1_l), o_A_r2_l = opaque_s(P.o_A_r2_l), o_A_r3_l = opaque_s(P.o_A_r3_l);
    const bool valid = wv.col < n_valid;
    const int e = e0 + (valid ? wv.col : 0);
    int src = 0, dst = 0, pose = 0;
    if constexpr (MODE == 1) {
        if (geo.ok) { src = geo.src; dst = geo.dst; } else { src = P.edge_src[e]; dst = P.edge_dst[e]; }
        pose = dst / P.nQ;
    } else if constexpr (MODE < 2) { src = P.edge_src[e]; dst = P.edge_dst[e]; pose = dst / P.nQ; }
    int nsrc = 0, ndst = 0;            // MODE 1: the next tile's indices, requested now
    if constexpr (MODE == 1) { if (e_next >= 0) { nsrc = P.edge_src[e_next]; ndst = P.edge_dst[e_next]; } }

    // operands of the edge pre-linear (first K-chunks) and its per-pose bias rows: requested before the geometry / length-encoding
    // VALU work, which hides their latency
    constexpr int NH = F0 / 32;      // pre-linear width: 128 (length + time embedding) or 64 (EBM critic: length only)
    constexpr int NT1 = H1 / 32, NT2 = H2 / 32;
    f32x16 h[NH];
    const int oA_pre = opaque_s(P.o_A_pre + scale * (NH * 4 * 256)), oAl_pre = opaque_s(P.o_A_pre_l + scale * (NH * 4 * 256));
    DenseRing<NH, 2> ring_pre{};
    DenseRing<NT1, 2> ring_r1u{};
    auto front_requests = [&]() {
        if constexpr (!UN) {
            const Buf tbb = make_buf(P.tb, P.tb_bytes);
            const int tvoff = (pose * P.tb_pose_stride) * 4 + wv.hi64;
            static_for<NH>([&]<int To>() { h[To] = ldrows(tbb, tvoff, scale * F0, To); });
            ring_pre = dense_prefetch<NH, 4, 2, HP>(wv, oA_pre, oAl_pre);
        } else ring_r1u = dense_prefetch<NT1, 4, 2, HP>(wv, o_A_r1, o_A_r1_l);
    };
    if constexpr (MODE != 1) { front_requests(); sched_fence(); }      // (MODE 1 asks for them only when a tile falls back to the per-edge front)

    // ---- geometry (graph_parser.py:159-215) ---------------------------------------------------------------------
    float vx = 0.0f, vy = 0.0f, vz = 1.0f, len;
    if constexpr (MODE == 2) len = (float)(e - 1) * P.rtab_step[scale];
    else if constexpr (MODE == 3) len = ((float)e + 0.5f) * P.rtab_step[scale];
    else if (MODE == 1 && geo.ok) { vx = geo.vx; vy = geo.vy; vz = geo.vz; len = sqrtf(vx * vx + vy * vy + vz * vz); }
    else {
        vx = P.key_x[3 * src + 0] - P.qpos[3 * dst + 0];
        vy = P.key_x[3 * src + 1] - P.qpos[3 * dst + 1];
        vz = P.key_x[3 * src + 2] - P.qpos[3 * dst + 2];
        len = sqrtf(vx * vx + vy * vy + vz * vz);
    }
    const float radius = P.radius[scale];
    float logit0 = 0.0f;
    if (!UN && radius > 0.0f) {
        const float cut = 1.0f - soft_step((len - P.cut_begin[scale]) / P.cut_div[scale]);
        logit0 = logf(fmaxf(cut, 1e-12f));
    }
    // (UNet layer: no cut-off on the SH; what the edge frame still needs is the reference's SH of a zero-length edge -- 0 for l > 0, the normalised
    //  zero vector -- which the general form gets from its unit vector (0, 0, 0): duplicate points of a cloud)
    const float cns = UN ? ((SO2 && !(len > 0.0f)) ? 0.0f : 1.0f) : soft_step((len - P.ns_lo) / P.ns_div);
    SH<L> Y;
    {
        const float inv = 1.0f / fmaxf(len, 1e-12f);
        const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
        Y.y0[0] = 1.0f;
        const float s3 = 1.7320508075688772f, s5 = 2.23606797749979f;
        Y.y1[0] = s3 * ux * cns; Y.y1[1] = s3 * uy * cns; Y.y1[2] = s3 * uz * cns;
        if constexpr (L >= 2) {
            const float rho = ux * ux + uz * uz;
            Y.y2[0] = s5 * s3 * ux * uz * cns;
            Y.y2[1] = s5 * s3 * ux * uy * cns;
            Y.y2[2] = s5 * (uy * uy - 0.5f * rho) * cns;
            Y.y2[3] = s5 * s3 * uy * uz * cns;
            Y.y2[4] = s5 * s3 * 0.5f * (uz * uz - ux * ux) * cns;
        }
        if constexpr (L >= 3) {      // e3nn component normalisation, y polar (diffusion_edf_amd/so3.py::spherical_harmonics)
            const float a = ux * uz, b = 0.5f * (uz * uz - ux * ux), rho = ux * ux + uz * uz;
            const float c0 = 4.183300132670378f /* sqrt(35/2) */, c1 = 10.246950765959598f /* sqrt(105) */, c2 = 1.620185174601965f /* sqrt(21/8) */,
                        c3 = 1.3228756555322954f /* sqrt(7)/2 */;
            Y.y3[0] = c0 * (a * uz + b * ux) * cns;
            Y.y3[1] = c1 * a * uy * cns;
            Y.y3[2] = c2 * (4.0f * uy * uy - rho) * ux * cns;
            Y.y3[3] = c3 * uy * (2.0f * uy * uy - 3.0f * rho) * cns;
            Y.y3[4] = c2 * uz * (4.0f * uy * uy - rho) * cns;
            Y.y3[5] = c1 * b * uy * cns;
            Y.y3[6] = c0 * (b * uz - a * ux) * cns;
        }
        static_assert(L <= 3, "spherical harmonics up to l = 3");
    }
    // SO2: the rotation g = R_x(beta) R_y(gamma) that takes the edge direction to the polar axis y (diffusion_edf_amd/so2.py::frame_angles):
    // cos gamma = z / rho, sin gamma = -x / rho (gamma = 0 when rho = 0), cos beta = y, sin beta = -rho; multiples by the addition theorems.
    // A zero-length edge takes the identity (its non-scalar SH vanish with the cut-off, any frame serves the (l, 0, l) paths).
    Trig<L> tg{};
    if constexpr (SO2) {
        const float inv = 1.0f / fmaxf(len, 1e-12f);
        const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
        const float rho2 = ux * ux + uz * uz;
        const bool planar = rho2 > 0.0f;
        const float rho = sqrtf(rho2), ir = planar ? 1.0f / rho : 0.0f;
        const bool real = len > 0.0f;
        tg.cg[0] = planar ? uz * ir : 1.0f; tg.sg[0] = planar ? -ux * ir : 0.0f;
        tg.cb[0] = real ? uy : 1.0f; tg.sb[0] = real ? -rho : 0.0f;
        static_for<L - 1>([&]<int m>() {
            tg.cg[m + 1] = tg.cg[m] * tg.cg[0] - tg.sg[m] * tg.sg[0]; tg.sg[m + 1] = tg.sg[m] * tg.cg[0] + tg.cg[m] * tg.sg[0];
            tg.cb[m + 1] = tg.cb[m] * tg.cb[0] - tg.sb[m] * tg.sb[0]; tg.sb[m + 1] = tg.sb[m] * tg.cb[0] + tg.cb[m] * tg.sb[0];
        });
    }

    constexpr int SPW = park_phys_slots<L>(), NSLOT = SPW + 2;      // SPW: segment softmax weights / normalisers
    __shared__ f32x4 park[NSLOT * 64];         // (declared here: the table path stages its rows in it before the parking starts)
    f32x16 r2[NT2];
    f32x4 trow[MODE == 1 ? 4 : 1][8];      // MODE 1: this lane's halves of the four table rows around its length (16 * NT2 floats of the 32 are used:
                                           // the row stride stays 256 B for the narrow radial MLP too)
    float tw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool tab = false;          // wave-uniform: this tile takes the radial network's front from the table
    if constexpr (MODE == 1) {
        DEDF_STAMP(0);
        const float pos = len * P.rtab_inv_step[scale];
        const bool accurate = __builtin_bit_cast(float, P.rtab_err[scale]) <= P.rtab_err_bound[scale];      // (wave-uniform; NaN compares false)
        tab = accurate && __all(!valid || pos < (float)P.rtab_n[scale]) != 0;
        if (tab) {
            const float ps = valid ? pos : 0.0f;
            const int i0 = (int)ps;
            const float u = ps - (float)i0;
            // 4-point Lagrange weights for the nodes at -1, 0, 1, 2 (rows i0 .. i0 + 3)
            const float um = u - 1.0f, up = u + 1.0f, u2 = u - 2.0f;
            const float w0 = -(1.0f / 6) * u * um * u2, w1 = 0.5f * up * um * u2, w2 = -0.5f * up * u * u2, w3 = (1.0f / 6) * up * u * um;
            const Buf rtb = make_buf(P.rtab, P.rtab_bytes);
            const int rv = ((P.rtab_row0[scale] + i0) * 2 + hi) * 128;
            // the four rows are only REQUESTED here; they are combined in the prologue of the fused stage, under its operand requests
            static_for<4>([&]<int K>() { static_for<4 * NT2>([&]<int Q>() { trow[K][Q] = bld4(rtb, rv, K * 256 + Q * 16); }); });
            tw[0] = w0; tw[1] = w1; tw[2] = w2; tw[3] = w3;
        } else front_requests();
        DEDF_STAMP(1);
    }
    if (!tab) {
        // ---- length embedding: this lane's 32 of the 64 channels (k = s + 32*hi) ------------------------------------------
        float eb[32];
        {
            const f32x4* const enc = reinterpret_cast<const f32x4*>(FRONT ? rows + RL::enc : P.W + P.o_enc + scale * 192);       // this scale's constants (edge_enc_to_lds)
            if (radius > 0.0f) {           // GaussianRadialBasis, radial_func.py:208-227
                // UNet layer (GaussianRadialBasisLayerFiniteCutoff, radial_func.py:262-278): t = (len - offset) / (cutoff - offset); the host
                // passes cutoff - offset as `radius` and the offset as `cut_begin`
                const float t = UN ? (len - P.cut_begin[scale]) / radius : len / radius;
                static_for<2>([&]<int Hf>() {      // 16 channels at a time, stage by stage (see sigmoid_stage)
                    float z[16], wv16[16];
                    static_for<4>([&]<int G4>() {
                        constexpr int G = 4 * Hf + G4;
                        const f32x4 mu = enc[hi * 8 + G], is = enc[16 + hi * 8 + G], w = enc[32 + hi * 8 + G];
                        static_for<4>([&]<int J>() { z[4 * G4 + J] = (t - mu[J]) * is[J]; wv16[4 * G4 + J] = w[J]; });
                    });
                    static_for<16>([&]<int i>() { z[i] = -0.5f * (z[i] * z[i]); });
                    static_for<16>([&]<int i>() { z[i] = z[i] * 1.44269504088896340736f; });
    #if defined(__HIP_DEVICE_COMPILE__)
                    static_for<16>([&]<int i>() { z[i] = __builtin_amdgcn_exp2f(z[i]); });
    #else
                    static_for<16>([&]<int i>() { z[i] = exp2f(z[i]); });
    #endif
                    static_for<16>([&]<int i>() { eb[16 * Hf + i] = z[i] * wv16[i]; });
                });
                if constexpr (UN) {        // soft_square_cutoff(t, thr = 0.8, infinite = False), radial_func.py:24-29: fades IN over t in (0, 0.2)
                    const float x1 = 1.0f - t;
                    const float c = t > 0.5f ? 1.0f : 1.0f - soft_step((x1 - 0.8f) / (1.0f - 0.8f));
                    static_for<32>([&]<int i>() { eb[i] = eb[i] * c; });
                }
            } else {                       // SinusoidalPositionEmbeddings(n = 1000), radial_func.py:305-316
                const float x = len / P.len_enc_max_r * 1000.0f;
                static_for<8>([&]<int G>() {
                    const f32x4 fr = enc[G];
                    static_for<4>([&]<int J>() { eb[4 * G + J] = sin_or_cos(x * fr[J], hi); });
                });
            }
        }

        DEDF_STAMP(0);
        // ---- edge pre-linear + SiLU (multiscale_tensor_field.py:225-234); time part + bias arrive as per-pose rows --------
        DenseRing<NT1, 2> ring_r1 = ring_r1u;
        if constexpr (!UN) {
            dense_rot_h<NH, 4, 2, HP>(wv, oA_pre, oAl_pre, h, [&]<int c, int j>() { return eb[8 * c + j]; }, ring_pre);
            ring_r1 = dense_prefetch<NT1, F0 / 16, 2, HP>(wv, o_A_r1, o_A_r1_l);      // layer 1's first operands: under the SiLU below
            sched_fence();
            static_for<NH>([&]<int To>() {
                to_vgpr(h[To]);
                float y[16];
                static_for<16>([&]<int R>() { y[R] = h[To][R]; });
                silu_stage<16>(y);
                static_for<16>([&]<int R>() { h[To][R] = y[R]; });
            });
        }
        DEDF_STAMP(1);
        // ---- RadialProfile layers 1, 2 (equiformer/radial_func.py:11-60) ---------------------------------------------------
        f32x16 r1[NT1];
        static_for<NT1>([&]<int To>() { r1[To] = ldrows_lds(r_b1, hi, 0, To); });
        if constexpr (UN)        // no pre-linear: layer 1 reads the radial basis (K = 64 = four chunks of this lane's embedding values)
            dense_rot_h<NT1, 4, 2, HP>(wv, o_A_r1, o_A_r1_l, r1, [&]<int c, int j>() { return eb[8 * c + j]; }, ring_r1);
        else
            dense_rot_h<NT1, F0 / 16, 2, HP>(wv, o_A_r1, o_A_r1_l, r1, [&]<int c, int j>() { return h[c / 2][8 * (c % 2) + j]; }, ring_r1);
        DEDF_STAMP(2);
        auto ring_r2 = dense_prefetch<NT2, H1 / 16, 2, HP>(wv, o_A_r2, o_A_r2_l);      // layer 2's first operands: under the LayerNorm below
        sched_fence();
        static_for<NT1>([&]<int To>() { to_vgpr(r1[To]); });
        ln_silu<NT1, UN>(r1, wv, r_g1, r_be1, P.ln_inv_n[0], P.ln_pad[0]);
        DEDF_STAMP(3);
        static_for<NT2>([&]<int To>() { r2[To] = ldrows_lds(r_b2, hi, 0, To); });
        dense_rot_h<NT2, H1 / 16, 2, HP>(wv, o_A_r2, o_A_r2_l, r2, [&]<int c, int j>() { return r1[c / 2][8 * (c % 2) + j]; }, ring_r2);
        DEDF_STAMP(4);
    }
    if constexpr (MODE == 3) {      // accuracy check: exact activations at the interval midpoint against the interpolation of table rows e .. e + 3
        static_for<NT2>([&]<int To>() { to_vgpr(r2[To]); });
        ln_silu<NT2, false>(r2, wv, r_g2, r_be2);
        const Buf rtb = make_buf(P.rtab, P.rtab_bytes);
        const int rv = ((P.rtab_row0[scale] + e) * 2 + hi) * 128;
        // 4-point Lagrange weights at u = 1/2 for the nodes at -1, 0, 1, 2
        const float w4[4] = {-0.0625f, 0.5625f, 0.5625f, -0.0625f};
        float err = 0.0f;
        bool bad = false;
        static_for<4 * NT2>([&]<int Q>() {
            f32x4 t = bld4(rtb, rv, Q * 16) * w4[0];
            static_for<3>([&]<int K>() { t = t + bld4(rtb, rv, (K + 1) * 256 + Q * 16) * w4[K + 1]; });
            // (fmaxf drops a NaN operand: a NaN in the table or in the exact activation is tracked on its own and must read as "inaccurate")
            static_for<4>([&]<int J>() { const float d = fabsf(t[J] - r2[Q / 4][4 * (Q % 4) + J]); bad = bad || !(d == d); err = fmaxf(err, d); });
        });
        if (bad) err = __builtin_inff();
        if (!valid) err = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
        for (int o = 32; o >= 1; o >>= 1) err = fmaxf(err, __shfl_xor(err, o, 64));
        if (wv.lane == 0) atomicMax(P.rtab_err + scale, __builtin_bit_cast(unsigned, err));      // err >= 0: the bit patterns order like the values
#endif
        return;
    }
    if constexpr (MODE == 2) {      // generator: layer 2's activation, one table row per grid node, done
        static_for<NT2>([&]<int To>() { to_vgpr(r2[To]); });
        ln_silu<NT2, false>(r2, wv, r_g2, r_be2);
        if (valid) {
            float* const o = P.rtab_out + ((size_t)(P.rtab_row0[scale] + e) * 2 + hi) * 32;
            static_for<4 * NT2>([&]<int Q>() { st4(o + 4 * Q, f32x4{r2[Q / 4][4 * (Q % 4)], r2[Q / 4][4 * (Q % 4) + 1], r2[Q / 4][4 * (Q % 4) + 2], r2[Q / 4][4 * (Q % 4) + 3]}); });
        }
        return;
    }
    // (layer 2's LayerNorm + SiLU run in the prologue of the next stage, under its first operand requests)

    // ---- layer 3 (-> per-edge TP weights, one 32-row tile at a time) fused with DTP #1 and the lin / sep_alpha GEMMs ----
    // Chunks are walked grouped by output degree (dedf_net.h::dtp_pos_chunk): first all l3 = 0 chunks into acc0 (NR0 tiles:
    // lin scalars + gates | alpha), then l3 = 1 into acc1, then l3 = 2 into acc2 (one tile per m).
    // When a group is complete its activations (logits, Gate) are computed and the gated features parked in LDS (this
    // wave's private 30 KB) until the second depth-wise TP reads them.
    f32x16 acc0[NR0], acc1[3], acc2[5], acc3[7];
    const Buf msgb = make_buf(P.msg, P.msg_bytes);
    const Buf msgd = make_buf(P.msg_dst, P.msg_dst_bytes);      // UNet layer: the destination's message is added (block.py:155); QT: the pose's time row (0e only)
    static_assert(!(UN && QT), "a UNet layer has no query time encoding");
    // per-l1 lane offsets of the 4 message rows this lane owns inside an 8-row group
    const int srcx = src;
    const int mv0 = srcx * (D * 4) + hi * 16, mv1 = srcx * (D * 4) + hi * 48, mv2 = srcx * (D * 4) + hi * 80, mv3 = srcx * (D * 4) + hi * 112;
    const int dv0 = (QT ? pose * (P.qd_pose_stride * 4) : dst * (D * 4)) + hi * 16, dv1 = dst * (D * 4) + hi * 48, dv2 = dst * (D * 4) + hi * 80, dv3 = dst * (D * 4) + hi * 112;
    // LDS parking: the gated features wait here as READY-MADE B operands of the value GEMMs: one 16-byte slot per lane holds the
    // fp16 hi halves of a 16-channel chunk (8 accumulator registers), the next slot their fp16 residuals ([slot][lane][8 halves];
    // chunk q = dedf_net.h::park_slot(degree, component, chunk) -> slots 2q | 2q + 1), lane-private
    f32x4* const pk = park + wv.lane;
    auto park_chunk = [&]<int Q, bool PADZ = false>(const float (&v)[8]) {
        if constexpr (park_packed<L>(Q)) {      // registers 2, 3, 6, 7 are the zero padding of 8x3e: hi and lo of the other four in one slot
            const float v4[4] = {v[0], v[1], v[4], v[5]};
            pk[(park_phys<L>(Q)) * 64] = split4pk(v4);
        } else {
            HL sp;
            if constexpr (PADZ) sp = split8zx<HP>(v); else sp = split8x<HP>(v);
            pk[(park_phys<L>(Q)) * 64] = __builtin_bit_cast(f32x4, sp.hi);
            if constexpr (!HP) pk[(park_phys<L>(Q) + 1) * 64] = __builtin_bit_cast(f32x4, sp.lo);
        }
    };

    // Software pipeline over the WN/16 chunks (16 weight rows = half a weight tile).  Region C issues, in one scheduling
    // region so that hipcc interleaves them:  the source-message loads of chunk C+2,  the layer-3 MFMAs of the NEXT weight
    // tile,  the lane-local Clebsch-Gordan VALU work + hi/lo split of chunk C+1 (-> its B operands),  the activations of a
    // group completed by chunk C-1,  and the lin / sep_alpha MFMAs of chunk C (A operands through a ring, PDA items ahead).
#ifndef DEDF_PDA3
#define DEDF_PDA3 2      // lmax 3: operand ring of the lin / sep_alpha stream two items deep (3: 143.3 k, 2: 149.0 k, 1: 148.4 k pose-steps/s;
                         // 76 / 66 / 54 spilled registers -- profiles/r03p_pda_lmax3_ab.log)
#endif
#ifndef DEDF_R2S_LDS3
#define DEDF_R2S_LDS3 1
#endif
#ifndef DEDF_R2S_LDS2
#define DEDF_R2S_LDS2 0
#endif
#ifndef DEDF_PDA2
#define DEDF_PDA2 3
#endif
    // (round 4, after the debug hooks left the table-reading kernel with 27 spare registers: its two operand rings one step deeper each -- lin / alpha
    //  stream 6 items, value stream 3 -- k_edge 2.92 -> 2.85 ms at 2.37 M edges in three interleaved A/B rounds, profiles/r04h_edge_variants_ab.log;
    //  512 registers, no scratch.  The per-edge kernels and lmax 3 keep their depths: 6 / 3 spill there or change nothing, r04i_*)
#ifndef DEDF_PDA2_TAB
#define DEDF_PDA2_TAB 5      // (6 until the MFMAs of the fused stage were spaced out, SGB1 below: 6 then leaves 16 B of scratch for the same time, r04z_sgb_rings_ab.log)
#endif
#ifndef DEDF_V_PDA_TAB
#define DEDF_V_PDA_TAB 3
#endif
#ifndef DEDF_PDA_SO2
#define DEDF_PDA_SO2 3      // edge-frame form: 3 against 5 (-2 %), 2 / 4 / 7 no better (profiles/r05c_so2_variants_ab.log, r05e_so2_timing_ab.log)
#endif
    constexpr int NCHK = WN / 16, PDA = SO2 ? DEDF_PDA_SO2 : (L == 3 ? DEDF_PDA3 : ((L == 2 && MODE == 1 && F0 == 128) ? DEDF_PDA2_TAB : DEDF_PDA2));
    struct XOps { f32x4 x[2][2 * L + 1]; f32x4 xd[UN ? 2 : 1][UN ? 2 * L + 1 : (QT ? 2 : 1)]; };      // (QT: xd[0][run], 0e chunks only)
    auto load_X = [&]<int C>() {      // this lane's 2 x 4 source-message rows of the chunk (contiguous runs in the reference layout)
        XOps o{};
        if constexpr (C < NCHK) {
            constexpr PathInfo pi = dtp_pos_path<L, SO2>(C);
            constexpr int l1 = pi.l1, d1 = 2 * l1 + 1, u0 = dtp_pos_u0<L, SO2>(C);
            const int mv = l1 == 0 ? mv0 : (l1 == 1 ? mv1 : (l1 == 2 ? mv2 : mv3));
            static_for<2>([&]<int run>() { static_for<d1>([&]<int Q>() {
                o.x[run][Q] = bld4(msgb, mv, (blk_off(l1) + (u0 + 8 * run) * d1 + 4 * Q) * 4);
            }); });
            if constexpr (UN) {
                const int dv = l1 == 0 ? dv0 : (l1 == 1 ? dv1 : (l1 == 2 ? dv2 : dv3));
                static_for<2>([&]<int run>() { static_for<d1>([&]<int Q>() {
                    o.xd[run][Q] = bld4(msgd, dv, (blk_off(l1) + (u0 + 8 * run) * d1 + 4 * Q) * 4);
                }); });
            }
            if constexpr (QT && l1 == 0) static_for<2>([&]<int run>() { o.xd[0][run] = bld4(msgd, dv0, (u0 + 8 * run) * 4); });
        }
        return o;
    };
    // SO2: this lane's 2 x 4 source channels of the current (input degree, channel range) in the edge frame; rotated once, read by every path of that range
    float xrot[8][2 * L + 1];
    auto valu_chunk = [&]<int C>(const XOps& xo, const f32x16& wtile) {
        BOpsH<L> o{};
        if constexpr (C < NCHK) {
            constexpr PathInfo pi = dtp_pos_path<L, SO2>(C);
            constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d1 = 2 * l1 + 1, d3 = 2 * l3 + 1, c2 = C % 2;
            using Cg = CG<l1, l2, l3>;
            if constexpr (SO2 && l3 >= 1) {
                // edge frame: B_t = (w . c_t / c_ref . cut-off) x'[i_t] for the terms (k_t, i_t, c_t) of the path (dedf_tables.h::kSo2*); c_ref rides on the A slot
                constexpr bool newx = C == 0 || so2_group_of<L>(dtp_pos_l3<L, SO2>(C > 0 ? C - 1 : 0)) != so2_group_of<L>(l3) || !dtp_pos_same_x<L, SO2>(C, C - 1);
                if constexpr (newx) static_for<2>([&]<int run>() { static_for<4>([&]<int j>() {
                    if constexpr (!pad_reg<L, NW>(l1, j)) {
                        float v[d1];
                        static_for<d1>([&]<int m>() {
                            constexpr int el = j * d1 + m;
                            if constexpr (UN) v[m] = xo.x[run][el / 4][el % 4] + xo.xd[run][el / 4][el % 4];
                            else if constexpr (QT && l1 == 0) v[m] = xo.x[run][0][el] + xo.xd[0][run][el];
                            else v[m] = xo.x[run][el / 4][el % 4];
                        });
                        Rot<l1>::in(v, tg);
                        static_for<d1>([&]<int m>() { xrot[4 * run + j][m] = v[m]; });
                    }
                }); });
                constexpr int NTm = kSo2NT[l1][l2][l3];
                constexpr float ref = kSo2Ref[l1][l2][l3];
                float wr[NTm][8];
                static_for<NTm>([&]<int t>() {
                    constexpr float ratio = kSo2C[l1][l2][l3][t] / ref, ar = ratio < 0.0f ? -ratio : ratio;
                    constexpr int rp = [&]() { for (int q = 0; q < t; ++q) { const float rq = kSo2C[l1][l2][l3][q] / ref; if ((rq < 0.0f ? -rq : rq) == ar) return q; } return t; }();
                    if constexpr (rp != t) static_for<8>([&]<int jj>() { wr[t][jj] = wr[rp][jj]; });
                    else if constexpr (l2 == 0 && ar == 1.0f) static_for<8>([&]<int jj>() { wr[t][jj] = wtile[8 * c2 + jj]; });
                    else {
                        const float kk = l2 == 0 ? ar : ar * cns;
                        static_for<8>([&]<int jj>() { wr[t][jj] = wtile[8 * c2 + jj] * kk; });
                    }
                    constexpr int I = kSo2I[l1][l2][l3][t];
                    float v[8];
                    static_for<8>([&]<int jj>() {
                        if constexpr (pad_reg<L, NW>(l1, jj % 4)) v[jj] = 0.0f;
                        else if constexpr (ratio < 0.0f) v[jj] = -(wr[t][jj] * xrot[jj][I]);
                        else v[jj] = wr[t][jj] * xrot[jj][I];
                    });
                    HL sp;
                    if constexpr (pad_reg<L, NW>(l1, 2)) sp = split8zx<HP>(v); else sp = split8x<HP>(v);
                    o.hi[t] = sp.hi; o.lo[t] = sp.lo;
                });
            } else
            if constexpr (dtp_pos_out<L, SO2>(C)) {      // output-side: B_i = w x_i, no contraction here
                float v[d1][8];
                static_for<2>([&]<int run>() {
                    static_for<d1>([&]<int Q>() { static_for<4>([&]<int i>() {
                        // element 4 Q + i of the run = component (4 Q + i) % d1 of channel (4 Q + i) / d1
                        constexpr int ch = (4 * Q + i) / d1, cmp = (4 * Q + i) % d1;
                        if constexpr (pad_reg<L, NW>(l1, ch)) v[cmp][4 * run + ch] = 0.0f;      // a zero-padding channel of the source rows
                        else {
                        float x;
                        if constexpr (UN) x = xo.x[run][Q][i] + xo.xd[run][Q][i];
                        else if constexpr (QT && l1 == 0) x = xo.x[run][Q][i] + xo.xd[0][run][i];
                        else x = xo.x[run][Q][i];
                        v[cmp][4 * run + ch] = x * wtile[8 * c2 + 4 * run + ch];
                        }
                    }); });
                });
                static_for<d1>([&]<int I>() { HL sp; if constexpr (pad_reg<L, NW>(l1, 2)) sp = split8zx<HP>(v[I]); else sp = split8x<HP>(v[I]); o.hi[I] = sp.hi; o.lo[I] = sp.lo; });
            } else {
            float m[Cg::NM];
            Cg::make(Y.template get<l2>(), m);
            float v[d3][8];
            static_for<2>([&]<int run>() {
                float xr[4 * d1];
                static_for<d1>([&]<int Q>() { static_for<4>([&]<int i>() {
                    if constexpr (UN) xr[4 * Q + i] = xo.x[run][Q][i] + xo.xd[run][Q][i];
                    else if constexpr (QT && l1 == 0) xr[4 * Q + i] = xo.x[run][Q][i] + xo.xd[0][run][i];
                    else xr[4 * Q + i] = xo.x[run][Q][i];
                }); });
                static_for<4>([&]<int j>() {
                    if constexpr (pad_reg<L, NW>(l1, j)) static_for<d3>([&]<int K>() { v[K][4 * run + j] = 0.0f; });      // a zero-padding channel of the source rows
                    else {
                    float t[d3];
                    Cg::apply(&xr[j * d1], m, t);
                    static_for<d3>([&]<int K>() { v[K][4 * run + j] = t[K] * wtile[8 * c2 + 4 * run + j]; });
                    }
                });
            });
            split_chunk<L, l3, pad_reg<L, NW>(l1, 2), HP>(v, o);
            }
        }
        return o;
    };
    const int o_S_lin = opaque_s(P.o_S_lin);
    AItem ring[PDA];
    static_for<PDA>([&]<int I>() { ring[I] = load_item<L, NR0, I, HP, SO2>(wv, o_S_lin); });
    // layer 3 on split-fp16 MFMAs: r2 (H2 rows = KC chunks) is split once per edge tile; per weight tile KC chunks x 3 MFMAs.
    // A operands (hi and lo image) form one global stream over all tiles.
    constexpr int KC = H2 / 16;
    HL r2s[KC];
    // lmax 3: the split layer-2 activations (the B operands of all 28 layer-3 weight tiles: 32 registers for the whole stage) wait in LDS instead,
    // in the slots of the l = 3 chunks and of the segment weights, which are only written when the last weight tile is done (finish_group<3>
    // runs in the last region, the layer-3 halves end three regions earlier)
    constexpr bool R2S_LDS = ((L == 3 && DEDF_R2S_LDS3) || (L == 2 && DEDF_R2S_LDS2)) && 2 * KC <= park_phys_slots<L>() + 2 - 2 * park_slot<L>(L, 0, 0);
    constexpr int R2S_SLOT = 2 * park_slot<L>(L, 0, 0);
    // Layer-3 work unit = half a weight tile (2 of the 4 K-chunks, 6 MFMAs).  Half P = 2 T + half of tile T runs in pipeline
    // region P - 3 into wbuf[T % 2]; its operands (and the tile's offset rows, the accumulator init) are requested one region
    // earlier, across a scheduling fence, so that the request cannot sink next to its use.  With the 32-wide MLP a tile has
    // only two K-chunks: its even half just seeds the accumulator, the odd half carries both chunks.
    struct L3Half { f32x4 h[2], l[2]; };
    auto l3_nk = []<int Ph>() { return KC == 4 ? 2 : (Ph % 2 ? 2 : 0); };
    auto load_l3 = [&]<int Ph>() {
        L3Half o{};
        if constexpr (Ph < 2 * NWT) static_for<l3_nk.template operator()<Ph>()>([&]<int k>() {
            constexpr int unit = KC == 4 ? 2 * Ph + k : 2 * (Ph / 2) + k;
            o.h[k] = bldw(wv, wv.lane16, (o_A_r3 + unit * 256) * 4);
            if constexpr (!HP) o.l[k] = bldw(wv, wv.lane16, (o_A_r3_l + unit * 256) * 4);
        });
        return o;
    };
    const int o_off_r3 = opaque_s(P.o_off_r3);
    auto load_off = [&]<int T>() {
        f32x16 o{};
        if constexpr (T < NWT) { if constexpr (RL::BIG) o = ldrows_lds(rows, hi, RL::off3, T); else o = ldrows(wv, o_off_r3, T); }
        return o;
    };
    auto run_l3 = [&]<int Ph>(const L3Half& a, const f32x16& init, f32x16& w) {
        if constexpr (Ph < 2 * NWT) {
            constexpr int c0 = KC == 4 ? 2 * (Ph % 2) : 0;
            f32x16 t = init;
            static_for<l3_nk.template operator()<Ph>()>([&]<int k>() {
                const h8 ah = __builtin_bit_cast(h8, a.h[k]), al = __builtin_bit_cast(h8, a.l[k]);
                HL b;
                if constexpr (R2S_LDS) {
                    b.hi = __builtin_bit_cast(h8, pk[(R2S_SLOT + 2 * (c0 + k)) * 64]);
                    if constexpr (!HP) b.lo = __builtin_bit_cast(h8, pk[(R2S_SLOT + 2 * (c0 + k) + 1) * 64]);
                } else b = r2s[c0 + k];
                t = mfma_h(ah, b.hi, t);
                if constexpr (!HP) { t = mfma_h(ah, b.lo, t); t = mfma_h(al, b.hi, t); }
            });
            w = t;
            if constexpr (Ph % 2 == 1) to_vgpr(w);      // finished tile: the VALU stage reads it
        }
    };
    // Debug hooks (stage tests): compiled into the per-edge instantiations only.  The table-reading kernel (MODE 1) is never launched in debug mode
    // (dedf_api.hip: use_tab requires !debug); there the hooks were 15 exec-masked store blocks in the hot loop and 4 in the value stage, each a
    // basic-block boundary in the middle of a pipeline region.
    constexpr bool DBG = MODE != 1 && !UN;
    auto dump_w = [&]<int Tw>(const f32x16& w) {      // debug only: back to the e3nn weight order
        if constexpr (DBG) if (P.dbg_w != nullptr && valid)
            static_for<16>([&]<int R>() {
                if constexpr (Tw * 32 + (R & 3) + 8 * (R >> 2) < WN) P.dbg_w[(size_t)e * WN + dtp_walk_row<L, SO2>(Tw * 32 + (R & 3) + 8 * (R >> 2)) + 4 * hi] = w[R] * P.w_unscale;
            });
    };

    // Output-side paths (dedf_net.h::dtp_path_out_side): G tiles of the path in flight, VALU-side sums of the contracted results per
    // output degree, and the contraction of a completed path  vacc[k] += (sum_j C_ijk Y_j) G_i  (one region after its last MFMAs).
    f32x16 go[3], gfin[3];
    float vacc1[3][16], vacc2[5][8], vacc3[7][8];
    auto contract_out = [&]<int Ce>(f32x16 (&G)[3]) {
        constexpr PathInfo pi = dtp_pos_path<L>(Ce);
        constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d1 = 2 * l1 + 1, d3 = 2 * l3 + 1;
        constexpr int NR = mul_of(l3) >= 32 ? 16 : mul_of(l3) / 2;
        using Cg = CG<l1, l2, l3>;
        float m[Cg::NM];
        Cg::make(Y.template get<l2>(), m);
        static_for<NR>([&]<int R>() {
            if constexpr (pad_reg<L, NW>(l3, R)) {      // a zero-padding output channel: nothing to contract
                if constexpr (dtp_pos_opens_vacc<L>(Ce)) static_for<d3>([&]<int K>() {
                    if constexpr (l3 == 1) vacc1[K][R] = 0.0f; else if constexpr (l3 == 2) vacc2[K][R] = 0.0f; else if constexpr (l3 == 3) vacc3[K][R] = 0.0f;
                });
            } else {
            float o[d3];
            static_for<d3>([&]<int K>() {
                if constexpr (dtp_pos_opens_vacc<L>(Ce)) o[K] = 0.0f;
                else if constexpr (l3 == 1) o[K] = vacc1[K][R];
                else if constexpr (l3 == 2) o[K] = vacc2[K][R];
                else o[K] = vacc3[K][R];
            });
            if constexpr (acc_paired<L>(l3)) {      // (two G tiles: mfma_chunk)
                if constexpr (d1 == 1) Cg::template acc<0>(HP ? G[0][R] : (G[0][R] + G[0][8 + R]) + G[1][R], m, o);
                else static_for<d1>([&]<int I>() { Cg::template acc<I>(I == 2 ? G[1][R] : G[0][8 * I + R], m, o); });
            } else
            if constexpr (d1 == 1) Cg::template acc<0>(HP ? G[0][R] : (G[0][R] + G[1][R]) + G[2][R], m, o);      // the three product terms
            else static_for<d1>([&]<int I>() { Cg::template acc<I>(G[I][R], m, o); });
            static_for<d3>([&]<int K>() {
                opaque_v(o[K]);      // accumulate here (see the value stage)
                if constexpr (l3 == 1) vacc1[K][R] = o[K]; else if constexpr (l3 == 2) vacc2[K][R] = o[K]; else vacc3[K][R] = o[K];
            });
            }
        });
    };

    // activations of a completed group ----------------------------------------------------------------------------------
    // (the accumulators carry the power-of-two operand scales: c_lin brings them back, the parked features carry u_scale)
    float logit[kHeads], g1[16], g2[8], g3[8];
    const float cl0 = opaque_s(P.c_lin[0]), us = opaque_s(P.u_scale), cl1 = opaque_s(P.c_lin[1]), cl2 = opaque_s(P.c_lin[L >= 2 ? 2 : 0]), cl3 = opaque_s(P.c_lin[L >= 3 ? 3 : 0]);
    auto finish_group = [&]<int l3>() {
        if constexpr (l3 == 0) {
            // attention logits (graph_attention.py:233-246): heads of sep_alpha -> SmoothLeakyReLU -> . alpha_dot + log cut-off
            constexpr int AT = alpha_row0<L>() / 32;
            static_for<kHeads>([&]<int hd>() {
                constexpr int T = AT + (hd >> 1), r0 = 8 * (hd & 1);
                const f32x4* adp = reinterpret_cast<const f32x4*>(rows + RL::adot + (hd >> 1) * 32 + hi * 16 + r0);
                const f32x4 d0 = adp[0], d1v = adp[1];
                // SmoothLeakyReLU(0.2) x normalize2mom (dedf_dev.h::slrelu_n), eight values stage by stage
                float x8[8];
                static_for<8>([&]<int R>() { x8[R] = acc0[T][r0 + R] * cl0; });
                on_live.template operator()<0>(x8, [&]<int N>(float (&x)[N]) {      // x <- slrelu_n(x); padded alpha rows stay 0
                    float sg[N];
                    sigmoid_stage<N>(x, sg);
                    static_for<N>([&]<int R>() { x[R] = (0.6f * x[R] + 0.4f * x[R] * (2.0f * sg[R] - 1.0f)) * kNormSlrelu; });
                });
                float sum = 0.0f;
                static_for<8>([&]<int R>() { if constexpr (!pad_reg<L, NW>(0, R)) sum += x8[R] * (R < 4 ? d0[R & 3] : d1v[R & 3]); });
                sum += xor32(sum);
                logit[hd] = sum + logit0;
            });
            // Gate (fast_activation.py:210-224): SiLU on the 64 scalars, sigmoid gates for the l >= 1 channels
            static_for<2>([&]<int T>() { static_for<2>([&]<int hf>() {
                float v[8];
                static_for<8>([&]<int J>() { v[J] = acc0[T][8 * hf + J] * cl0; });
                on_live.template operator()<0>(v, [&]<int N>(float (&x)[N]) { silu_stage<N>(x); static_for<N>([&]<int J>() { x[J] = x[J] * (kNormSilu * us); }); });
                park_chunk.template operator()<park_slot<L>(0, 0, 2 * T + hf), pad_reg<L, NW>(0, 2)>(v);
            }); });
            if constexpr (L >= 1) {
                constexpr int G0 = gate_row(1, 0);
                static_for<16>([&]<int R>() { g1[R] = acc0[G0 / 32][(G0 % 32) / 2 + R] * cl0; });
                on_live.template operator()<1>(g1, [&]<int N>(float (&x)[N]) { float sg[N]; sigmoid_stage<N>(x, sg); static_for<N>([&]<int R>() { x[R] = sg[R] * (kNormSigmoid * (cl1 * us)); }); });
            }
            if constexpr (L >= 2) {
                constexpr int G0 = gate_row(2, 0);
                static_for<8>([&]<int R>() { g2[R] = acc0[G0 / 32][(G0 % 32) / 2 + R] * cl0; });
                on_live.template operator()<2>(g2, [&]<int N>(float (&x)[N]) { float sg[N]; sigmoid_stage<N>(x, sg); static_for<N>([&]<int R>() { x[R] = sg[R] * (kNormSigmoid * (cl2 * us)); }); });
            }
            if constexpr (L >= 3) {
                constexpr int G0 = gate_row(3, 0);
                static_for<8>([&]<int R>() { g3[R] = acc0[G0 / 32][(G0 % 32) / 2 + R] * cl0; });
                on_live.template operator()<3>(g3, [&]<int N>(float (&x)[N]) { float sg[N]; sigmoid_stage<N>(x, sg); static_for<N>([&]<int R>() { x[R] = sg[R] * (kNormSigmoid * (cl3 * us)); }); });
            }
        } else if constexpr (l3 == 1) {
            static_for<3>([&]<int K>() { static_for<2>([&]<int hf>() {
                float v[8];
                static_for<8>([&]<int J>() {
                    if constexpr (pad_reg<L, NW>(1, J)) v[J] = 0.0f;
                    else if constexpr (dtp_group_has_out<L, SO2>(1)) v[J] = (acc1[K][8 * hf + J] + vacc1[K][8 * hf + J]) * g1[8 * hf + J];
                    else v[J] = acc1[K][8 * hf + J] * g1[8 * hf + J];
                });
                park_chunk.template operator()<park_slot<L>(1, K, hf), pad_reg<L, NW>(1, 2)>(v);
            }); });
        } else if constexpr (l3 == 2) {
            static_for<5>([&]<int K>() {      // 16 channels = registers 0-7
                float v[8];
                static_for<8>([&]<int J>() {
                    constexpr int T = acc_paired<L>(2) ? K / 2 : K, R = acc_paired<L>(2) ? 8 * (K % 2) + J : J;      // (paired: see mfma_chunk)
                    if constexpr (pad_reg<L, NW>(2, J)) v[J] = 0.0f;
                    else if constexpr (dtp_group_has_out<L, SO2>(2)) v[J] = (acc2[T][R] + vacc2[K][J]) * g2[J]; else v[J] = acc2[T][R] * g2[J];
                });
                park_chunk.template operator()<park_slot<L>(2, K, 0), pad_reg<L, NW>(2, 2)>(v);
            });
        } else {
            static_for<7>([&]<int K>() {      // 16 channels (8 of them the zero padding of 8x3e) = registers 0-7
                float v[8];
                static_for<8>([&]<int J>() {
                    constexpr int T = acc_paired<L>(3) ? K / 2 : K, R = acc_paired<L>(3) ? 8 * (K % 2) + J : J;
                    if constexpr (pad_reg<L, NW>(3, J)) v[J] = 0.0f;
                    else if constexpr (dtp_group_has_out<L, SO2>(3)) v[J] = (acc3[T][R] + vacc3[K][J]) * g3[J]; else v[J] = acc3[T][R] * g3[J];
                });
                park_chunk.template operator()<park_slot<L>(3, K, 0)>(v);
            });
        }
    };
    auto start_group = [&]<int l3>() {
        if constexpr (l3 == 1) static_for<3>([&]<int K>() { static_for<16>([&]<int R>() { acc1[K][R] = 0.0f; }); });
        if constexpr (l3 == 2) static_for<acc_paired<L>(2) ? 3 : 5>([&]<int K>() { static_for<16>([&]<int R>() { acc2[K][R] = 0.0f; }); });
        if constexpr (l3 == 3) static_for<acc_paired<L>(3) ? 4 : 7>([&]<int K>() { static_for<16>([&]<int R>() { acc3[K][R] = 0.0f; }); });
    };

    DEDF_STAMP(6);
    // prologue: weight tile 0 and the first half of tile 1, source rows of chunks 0 / 1, B operands of chunk 0
    f32x16 wbuf[2];
    XOps x_nxt = load_X.template operator()<1>();
    BOpsH<L> b_cur;
    L3Half l3n;
    f32x16 offn;
    {
        const XOps x0 = load_X.template operator()<0>();
        const L3Half p0 = load_l3.template operator()<0>(), p1 = load_l3.template operator()<1>(), p2 = load_l3.template operator()<2>();
        const f32x16 off0 = load_off.template operator()<0>(), off1 = load_off.template operator()<1>();
        sched_fence();
        // layer 2's LayerNorm + SiLU and the split of its output (B operands of layer 3), under the requests above
        if (!tab) {
            static_for<NT2>([&]<int To>() { to_vgpr(r2[To]); });
            ln_silu<NT2, UN>(r2, wv, r_g2, r_be2, P.ln_inv_n[1], P.ln_pad[1]);
        } else if constexpr (MODE == 1) {
            static_for<4 * NT2>([&]<int Q>() { static_for<4>([&]<int J>() {
                r2[Q / 4][4 * (Q % 4) + J] = (tw[0] * trow[0][Q][J] + tw[1] * trow[1][Q][J]) + (tw[2] * trow[2][Q][J] + tw[3] * trow[3][Q][J]);
            }); });
        }
        DEDF_STAMP(5);
        static_for<KC>([&]<int c>() {
            float t[8];
            static_for<8>([&]<int J>() { t[J] = r2[c / 2][8 * (c % 2) + J]; });
            r2s[c] = split8x<HP>(t);
            if constexpr (R2S_LDS) {
                pk[(R2S_SLOT + 2 * c) * 64] = __builtin_bit_cast(f32x4, r2s[c].hi);
                if constexpr (!HP) pk[(R2S_SLOT + 2 * c + 1) * 64] = __builtin_bit_cast(f32x4, r2s[c].lo);
            }
        });
        static_for<NR0>([&]<int T>() {      // accumulator init: lin / sep_alpha biases
            if constexpr (RL::BIG) acc0[T] = ldrows_lds(rows, hi, RL::b0, T); else acc0[T] = ldrows(wv, P.o_b_r0, T);
        });
        sched_fence();
        l3n = load_l3.template operator()<3>();
        offn = load_off.template operator()<2>();
        sched_fence();
        run_l3.template operator()<0>(p0, off0, wbuf[0]);
        run_l3.template operator()<1>(p1, wbuf[0], wbuf[0]);
        run_l3.template operator()<2>(p2, off1, wbuf[1]);
        sched_fence();
        dump_w.template operator()<0>(wbuf[0]);
        b_cur = valu_chunk.template operator()<0>(x0, wbuf[0]);
    }
    DEDF_STAMP(7);
    static_for<NCHK>([&]<int C>() {
        constexpr int Ph = C + 3, T3 = Ph / 2;       // layer-3 half of this region
        XOps x_nn = x_nxt;                        // chunks that read the same source rows share the request
        if constexpr (!dtp_pos_same_x<L, SO2>(C + 2, C + 1)) x_nn = load_X.template operator()<C + 2>();
        const L3Half l3c = l3n;
        const f32x16 offc = offn;
        l3n = load_l3.template operator()<Ph + 1>();
        if constexpr ((Ph + 1) % 2 == 0) offn = load_off.template operator()<(Ph + 1) / 2>();
        // (SO2: every output degree >= 1 is live from the end of the scalar group to the end of the walk)
        static_for<L + 1>([&]<int g>() { if constexpr (g >= 1 && C == (SO2 ? so2_group_end<L>(so2_group_of<L>(g) - 1) : dtp_group_end<L>(g - 1))) start_group.template operator()<g>(); });
        sched_fence();
        if constexpr (Ph % 2 == 0) run_l3.template operator()<Ph>(l3c, offc, wbuf[T3 % 2]);
        else run_l3.template operator()<Ph>(l3c, wbuf[T3 % 2], wbuf[T3 % 2]);
        const BOpsH<L> b_nxt = valu_chunk.template operator()<C + 1>(x_nxt, wbuf[((C + 1) / 2) % 2]);
        constexpr bool fin_out = C >= 1 && dtp_pos_out<L, SO2>(C - 1) && dtp_pos_path_last<L, SO2>(C - 1);      // an output-side path ended at C - 1
        if constexpr (fin_out) static_for<acc_paired<L>(dtp_pos_l3<L, SO2>(C >= 1 ? C - 1 : 0)) ? 2 : 3>([&]<int a>() { gfin[a] = go[a]; });
        mfma_chunk<L, NR0, C, HP, PDA, SO2>(wv, o_S_lin, ring, b_cur, acc0, acc1, acc2, acc3, go);
        if constexpr (fin_out) contract_out.template operator()<C - 1>(gfin);
        static_for<L + 1>([&]<int g>() { if constexpr (C == (SO2 ? so2_group_end<L>(so2_group_of<L>(g)) : dtp_group_end<L>(g))) finish_group.template operator()<g>(); });
        // One MFMA per SGB1 VALU instructions inside the region.  hipcc's own schedule issues the region's 15-27 ready MFMAs in bursts, and a lone in-order
        // wave issues nothing while a burst drains (32 cycles per MFMA); spaced by the Clebsch-Gordan / split work of the next chunk they run under it.
        // Round 4, the sampler's two timed instantiations only (interleaved A/B at 2.37 M edges, profiles/r04z_sgb*_ab.log): lmax 2 2.697 -> 2.624 ms
        // with 5 (3 / 4: slower, 6 - 8: 2.64 - 2.68), lmax 3 5.53 -> 5.27 ms with 7 (5: 5.30, 9: 5.35, 12: 5.32); the per-edge lmax-2 kernel spills
        // 224 B with it, the value stage's regions gain nothing (their bursts have no VALU partner).
#ifndef DEDF_SGB_SO2
#define DEDF_SGB_SO2 5      // (table-reading kernels only, like the general form: 3 / 4 / 5 / 7 -> -1 / -2 / -2 / -2 %, profiles/r05c_so2_variants_ab.log)
#endif
        constexpr int SGB1 = SO2 ? (MODE == 1 ? DEDF_SGB_SO2 : 0) : ((MODE == 1 && F0 == 128 && H1 == 128 && H2 == 64 && !HP && !UN) ? (L == 3 ? 7 : (L == 2 ? 5 : 0)) : 0);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SGB1 > 0) static_for<28>([&]<int i>() { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, SGB1, 0); });
#endif
        sched_fence();
        if constexpr (Ph % 2 == 1 && T3 < NWT) dump_w.template operator()<T3>(wbuf[T3 % 2]);
        x_nxt = x_nn; b_cur = b_nxt;
        if constexpr (C + 1 == dtp_group_end<L>(0)) DEDF_STAMP(8);
        if constexpr (L >= 2 && C + 1 == dtp_group_end<L>(1)) DEDF_STAMP(9);
        if constexpr (L == 3 && C + 1 == dtp_group_end<L>(2)) DEDF_STAMP(15);
    });
    if constexpr (dtp_pos_out<L, SO2>(NCHK - 1)) contract_out.template operator()<NCHK - 1>(go);
    if constexpr (SO2) static_for<L>([&]<int g>() { if constexpr (so2_group_end<L>(so2_group_of<L>(g + 1)) == NCHK) finish_group.template operator()<g + 1>(); });
    else finish_group.template operator()<L>();
    sched_fence();
    DEDF_STAMP(12);
    float nk[3] = {0.0f, 0.0f, 0.0f}, nq[3] = {0.0f, 0.0f, 0.0f};       // MODE 1: the next tile's coordinates, requested here
    if constexpr (MODE == 1) {
        if (e_next >= 0) {
            static_for<3>([&]<int i>() { nk[i] = P.key_x[3 * nsrc + i]; nq[i] = P.qpos[3 * ndst + i]; });
        }
        sched_fence();
    }
    // ---- sep_value: depth-wise TP #2 (shared weights folded into the A stream) + LinearRS -> value --------------------------
    // The Clebsch-Gordan coefficient products of the SH (CG::make) are recomputed here from an opaque copy of the SH: merged with the
    // first depth-wise TP's, ~130 of them would stay alive across both phases.
    static_for<L + 1>([&]<int l>() { static_for<2 * l + 1>([&]<int i>() {
        if constexpr (l == 0) opaque_v(Y.y0[i]); else if constexpr (l == 1) opaque_v(Y.y1[i]); else if constexpr (l == 2) opaque_v(Y.y2[i]); else opaque_v(Y.y3[i]);
    }); });
    // Output-side form (dedf_net.h::make_val_walk): the parked gated features are the B operands as they are, the GEMM results
    // G_i (one tile per component i of the input degree) are contracted with the edge's SH on the VALU into val0 / val1[m] /
    // val2[m]; paths are walked by output degree, a completed degree goes straight to the segment record.
    // (plain scalars, not 16-register tuples: the VALU updates them element by element)
    float val0[2][16], val1[3][16], val2[5][8], val3[7][8];
    if constexpr (!SO2) static_for<2>([&]<int T>() { const f32x16 b = ldrows_lds(rows, hi, RL::val0, T); static_for<16>([&]<int R>() { val0[T][R] = b[R]; }); });
    // ---- joint-softmax partials --------------------------------------------------------------------------------------------
    // The tile's edges are ordered by destination, so the edges of one destination form a run of lanes ("segment").  Instead
    // of one 976-byte record per edge, the tile emits one per segment: the segment's softmax-weighted mean value and the
    // log-sum-exp of its logits -- k_aggregate merges them exactly as it would merge edges (softmax of softmaxes).  Segmented
    // scans over the 32 edge columns of each half-wave run on ds_bpermute lane shuffles (no LDS memory involved).
    const float cv0 = opaque_s(P.c_val[0]), cv1 = opaque_s(P.c_val[1]), cv2 = opaque_s(P.c_val[L >= 2 ? 2 : 0]), cv3 = opaque_s(P.c_val[L >= 3 ? 3 : 0]);
    // byte addresses of lane - s / lane + s (s = 1, 2, 4, 8, 16) are formed where they are used (ds_bpermute reads bits 7:2)
    const int lane4 = wv.lane * 4;
    auto sh_up_a = [&](int i) { return lane4 - (4 << i); };
    auto sh_dn_a = [&](int i) { return lane4 + (4 << i); };
    auto shf = [&](int addr, float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x))); };
    auto shi = [&](int addr, int x) { return __builtin_amdgcn_ds_bpermute(addr, x); };
    const int col = wv.col;
    int seg_start;                              // column of the first edge of this lane's segment
    bool seg_last;                              // this lane is the last edge of its segment (and a real edge): it stores the record
    // scans over the 32 columns of a half-wave on DPP: row_shr / row_shl inside the 16-lane rows, row_bcast:15 to carry the lower
    // row's last lane into the upper row (rows 1 and 3 only)
    auto dpi = [&]<int CTRL, int ROWS, bool ZERO>(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWS, 0xf, ZERO); };
    auto dpf = [&]<int CTRL, int ROWS, bool ZERO>(float old, float x) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, ZERO));
    };
    auto mkr = [&](int i) { return col - (1 << i) >= max(seg_start, col & 16); };      // lane - 2^i is in this lane's row and segment
    {
        const int prev_dst = shi(sh_up_a(0), dst);
        const bool head = col == 0 || dst != prev_dst || col == n_valid;      // padding lanes (col >= n_valid) form their own segment
        seg_start = head ? col : 0;             // inclusive prefix maximum over the columns
        static_for<4>([&]<int i>() { seg_start = max(seg_start, dpi.template operator()<0x110 + (1 << i), 0xf, true>(0, seg_start)); });
        seg_start = max(seg_start, dpi.template operator()<0x142, 0xa, false>(0, seg_start));
        const int next_head = shi(sh_dn_a(0), head ? 1 : 0);
        const bool last_any = col == 31 || next_head != 0;
        seg_last = last_any && valid;
        int seg_end = last_any ? col : 31;      // inclusive suffix minimum: column of the segment's last edge
        static_for<4>([&]<int i>() { seg_end = min(seg_end, dpi.template operator()<0x100 + (1 << i), 0xf, false>(31, seg_end)); });
        { const int up = shi(((wv.lane & 32) + 16) * 4, seg_end); if (col < 16) seg_end = min(seg_end, up); }
        const int end_addr = ((wv.lane & 32) + seg_end) * 4;
        const bool crossb = col >= 16 && seg_start < 16;
        float lse[kHeads], pw[kHeads], inv_s[kHeads];
        static_for<kHeads>([&]<int h>() {
            float m = logit[h];                  // inclusive prefix maximum along the segment, then the value at its last lane
            static_for<4>([&]<int i>() { const float t = dpf.template operator()<0x110 + (1 << i), 0xf, false>(m, m); m = mkr(i) ? fmaxf(m, t) : m; });
            { const float t = dpf.template operator()<0x142, 0xa, false>(m, m); m = crossb ? fmaxf(m, t) : m; }
            m = shf(end_addr, m);
            pw[h] = valid ? fexp(logit[h] - m) : 0.0f;
            float sum = pw[h];
            static_for<4>([&]<int i>() { const float t = dpf.template operator()<0x110 + (1 << i), 0xf, true>(0.0f, sum); sum += mkr(i) ? t : 0.0f; });
            { const float t = dpf.template operator()<0x142, 0xa, false>(0.0f, sum); sum += crossb ? t : 0.0f; }
            inv_s[h] = 1.0f / sum;               // meaningful on the segment's last lane only
            lse[h] = m + logf(sum);
        });
        if (seg_last && hi == 0) st4(P.out + (size_t)(e0 + seg_start) * REC + D, f32x4{lse[0], lse[1], lse[2], lse[3]});
        // the per-head weights wait in LDS until each irreps block is emitted (8 registers less during the value GEMMs)
        // point attention (gnn_block.py:190-194, graph_attention.py:257-258): the edge's softmax weight is multiplied by its key
        // point's weight AFTER the normalisation, i.e. the values are weighted by pw * w_src while the sums above are not
        const float wsrc = P.key_w != nullptr ? P.key_w[src] : 1.0f;
        pk[SPW * 64] = f32x4{pw[0], pw[1], pw[2], pw[3]} * wsrc;
        pk[(SPW + 1) * 64] = f32x4{inv_s[0], inv_s[1], inv_s[2], inv_s[3]};
    }
    auto orec_of = [&]() { return P.out + (size_t)(e0 + seg_start) * REC; };
    auto drec_of = [&]() -> float* { if constexpr (DBG) return P.dbg_out != nullptr ? P.dbg_out + (size_t)e * REC : nullptr; else return nullptr; };
    // The 240 values are reduced with DPP row shifts (VALU only): an inclusive segmented scan inside each 16-lane row
    // (row_shr 1, 2, 4, 8; sources outside the row read 0), then lane 15 of the lower row is added to the lanes of the upper
    // row whose segment began in the lower row (row_bcast:15).  Steps beyond the longest segment of the tile are skipped.
    int n_steps = 0;                            // wave-uniform: in-row steps some lane still needs
    static_for<4>([&]<int i>() { if (__builtin_amdgcn_ballot_w64(mkr(i)) != 0) n_steps = i + 1; });
    // x[q] += m * x[q] of the lane selected by the DPP control, four registers per asm block (one v_fmac_f32_dpp each; the
    // s_nop covers the VALU-write -> DPP-read wait states that hipcc cannot see inside the block)
#if defined(__HIP_DEVICE_COMPILE__)
#define DEDF_FMAC_DPP4(x, m, CTRL)                                                                                                  \
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 " CTRL "\n\tv_fmac_f32_dpp %1, %1, %4 " CTRL "\n\tv_fmac_f32_dpp %2, %2, %4 " CTRL \
                 "\n\tv_fmac_f32_dpp %3, %3, %4 " CTRL                                                                               \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(m))
#else
#define DEDF_FMAC_DPP4(x, m, CTRL) (void)(m)
#endif
    auto scan_step = [&]<int i>(float (&x)[4], float m) {
        if constexpr (i == 0) DEDF_FMAC_DPP4(x, m, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else if constexpr (i == 1) DEDF_FMAC_DPP4(x, m, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else if constexpr (i == 2) DEDF_FMAC_DPP4(x, m, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else if constexpr (i == 3) DEDF_FMAC_DPP4(x, m, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else DEDF_FMAC_DPP4(x, m, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    };
    // NS record slots of one irreps block at once (x[n] = this lane's four channel values of slot n, true units, already
    // weighted); afterwards the segment's last lane stores its means
    auto emit = [&]<int NS>(f32x4 (&xv)[NS], const int (&rec_off)[NS], const float (&inv)[NS]) {
        float x[NS][4];
        static_for<NS>([&]<int n>() { static_for<4>([&]<int q>() { x[n][q] = xv[n][q]; }); });
#ifndef DEDF_SCAN_UNCOND
#define DEDF_SCAN_UNCOND 0
#endif
        static_for<4>([&]<int i>() {
            // (SO2 && DEDF_SCAN_UNCOND: every step always -- a step beyond the longest segment adds 0 x the shifted value --, so that the piece is
            //  straight-line code the region's MFMAs can be spread over)
            if ((SO2 && DEDF_SCAN_UNCOND) || i < n_steps) {
                const float m = mkr(i) ? 1.0f : 0.0f;
                static_for<NS>([&]<int n>() { scan_step.template operator()<i>(x[n], m); });
            }
        });
        const float cross = (col >= 16 && seg_start < 16) ? 1.0f : 0.0f;
        static_for<NS>([&]<int n>() { scan_step.template operator()<4>(x[n], cross); });
        float* const orec = orec_of();
        if (seg_last) static_for<NS>([&]<int n>() { st4(orec + rec_off[n], f32x4{x[n][0], x[n][1], x[n][2], x[n][3]} * inv[n]); });
    };
    // (K0, NK: the components K0 .. K0 + NK - 1 of the block only -- output tiles for l3 = 0 --, NK < 0: all of them.  The edge-frame value stage
    //  emits a completed degree piece by piece under the GEMMs of the next one.)
    auto store_group = [&]<int l3, int K0 = 0, int NKsel = -1>() {       // value in internal layout [l][m][channel]; head of a channel = channel / (mul / 4)
        const f32x4 pwv = pk[SPW * 64], ivv = pk[(SPW + 1) * 64];
        const float pw[kHeads] = {pwv[0], pwv[1], pwv[2], pwv[3]}, inv_s[kHeads] = {ivv[0], ivv[1], ivv[2], ivv[3]};
        float* const drec = drec_of();
        constexpr int NKall = l3 == 0 ? 2 : 2 * l3 + 1, NK = NKsel < 0 ? NKall : NKsel;
        if constexpr (l3 == 0) {
            f32x4 x[4 * NK]; int ro[4 * NK]; float iv[4 * NK];
            static_for<NK>([&]<int Tq>() { static_for<4>([&]<int g>() {
                constexpr int T = K0 + Tq, hd = 2 * T + g / 2, n = 4 * Tq + g;
                ro[n] = T * 32 + 8 * g + 4 * hi; iv[n] = inv_s[hd];
                x[n] = f32x4{val0[T][4 * g], val0[T][4 * g + 1], val0[T][4 * g + 2], val0[T][4 * g + 3]};
                if constexpr (DBG) { x[n] = x[n] * cv0; if (drec != nullptr && valid) st4(drec + ro[n], x[n]); x[n] = x[n] * pw[hd]; }
                else {
                    const float sc_ = cv0 * pw[hd];
                    x[n] = x[n] * sc_;
                }          // (one product per value: the power-of-two operand scale folds into the softmax weight exactly)
            }); });
            emit(x, ro, iv);
        } else if constexpr (l3 == 1) {
            f32x4 x[4 * NK]; int ro[4 * NK]; float iv[4 * NK];
            static_for<NK>([&]<int Kq>() { static_for<4>([&]<int g>() {
                constexpr int K = K0 + Kq, n = 4 * Kq + g;
                ro[n] = blk_off(1) + K * mul_of(1) + 8 * g + 4 * hi; iv[n] = inv_s[g];
                x[n] = f32x4{val1[K][4 * g], val1[K][4 * g + 1], val1[K][4 * g + 2], val1[K][4 * g + 3]};
                if constexpr (DBG) { x[n] = x[n] * cv1; if (drec != nullptr && valid) st4(drec + ro[n], x[n]); x[n] = x[n] * pw[g]; }
                else {
                    const float sc_ = cv1 * pw[g];
                    x[n] = x[n] * sc_;
                }
            }); });
            emit(x, ro, iv);
        } else if constexpr (l3 == 2) {
            f32x4 x[2 * NK]; int ro[2 * NK]; float iv[2 * NK];
            static_for<NK>([&]<int Kq>() { static_for<2>([&]<int g>() {        // 16 channels: head = 2 g + hi
                constexpr int K = K0 + Kq, n = 2 * Kq + g;
                ro[n] = blk_off(2) + K * mul_of(2) + 8 * g + 4 * hi; iv[n] = hi ? inv_s[2 * g + 1] : inv_s[2 * g];
                x[n] = f32x4{val2[K][4 * g], val2[K][4 * g + 1], val2[K][4 * g + 2], val2[K][4 * g + 3]};
                if constexpr (DBG) { x[n] = x[n] * cv2; if (drec != nullptr && valid) st4(drec + ro[n], x[n]); x[n] = x[n] * (hi ? pw[2 * g + 1] : pw[2 * g]); }
                else {
                    const float sc_ = cv2 * (hi ? pw[2 * g + 1] : pw[2 * g]);
                    x[n] = x[n] * sc_;
                }
            }); });
            emit(x, ro, iv);
        } else {
            f32x4 x[2 * NK]; int ro[2 * NK]; float iv[2 * NK];
            static_for<NK>([&]<int Kq>() { static_for<2>([&]<int g>() {        // 16 (padded) channels: head = 2 g + hi
                constexpr int K = K0 + Kq, n = 2 * Kq + g;
                ro[n] = blk_off(3) + K * mul_of(3) + 8 * g + 4 * hi; iv[n] = hi ? inv_s[2 * g + 1] : inv_s[2 * g];
                x[n] = f32x4{val3[K][4 * g], val3[K][4 * g + 1], val3[K][4 * g + 2], val3[K][4 * g + 3]};
                if constexpr (DBG) { x[n] = x[n] * cv3; if (drec != nullptr && valid) st4(drec + ro[n], x[n]); x[n] = x[n] * (hi ? pw[2 * g + 1] : pw[2 * g]); }
                else {
                    const float sc_ = cv3 * (hi ? pw[2 * g + 1] : pw[2 * g]);
                    x[n] = x[n] * sc_;
                }
            }); });
            emit(x, ro, iv);
        }
    };
    if constexpr (SO2) {
    // ---- the value in the edge frame (dedf_net.h::make_sval_walk) ----------------------------------------------------------------------------
    // Every item is a run of plain GEMM terms: A slot (coefficient class folded in) x parked component (as it is, or with its sign bits flipped)
    // into the tile of its output component; set 0 collects the paths with l2 = 0, set 1 those that carry the non-scalar cut-off.  A completed
    // degree is read once, value' = set0 + cns . set1, rotated back to the global frame and handed to the segmented reduction.
#ifndef DEDF_V_PDA_SO2
#define DEDF_V_PDA_SO2 3
#endif
    constexpr int NVI = sval_num_items<L>(), PDV = DEDF_V_PDA_SO2, RS = 2 * (PDV + 1), NB = sval_max_ops<L>();
    const int o_S_val = opaque_s(P.o_S_val);
    struct ASlot { f32x4 h, l, hu, lu; };      // (hu, lu: the same 16 rows placed as rows 16-31 -- paired tiles, lmax 3)
    struct SB { f32x4 h[NB], l[NB]; };
    ASlot aring[RS];
    float tok = logit0;
    int lane16_t = wv.lane16, lane16_r16_t = wv.lane16_r16, lane16_r16up_t = wv.lane16_r16up, lane_t = wv.lane;
    auto load_A = [&]<int S>() {
        if constexpr (S < sval_num_slots<L>()) {
            const int lv = mul_of(sval_slot_l3<L>(S)) < 32 ? lane16_r16_t : lane16_t;      // rows 16-31 are padding: half the lanes fetch
            if constexpr (sval_slot_needs<L>(S, false)) {
                aring[S % RS].h = bldw(wv, lv, (o_S_val + S * 512) * 4);
                if constexpr (!HP) aring[S % RS].l = bldw(wv, lv, (o_S_val + S * 512 + 256) * 4);
            }
            if constexpr (sval_slot_needs<L>(S, true)) {
                aring[S % RS].hu = bldw(wv, lane16_r16up_t, (o_S_val + S * 512) * 4);
                if constexpr (!HP) aring[S % RS].lu = bldw(wv, lane16_r16up_t, (o_S_val + S * 512 + 256) * 4);
            }
        }
    };
    auto load_A_of = [&]<int I>() {
        if constexpr (I < NVI) static_for<sval_item<L>(I).new_slots>([&]<int n>() { load_A.template operator()<sval_item_slot0<L>(I) + n>(); });
    };
    auto load_B = [&]<int I>() {
        SB o{};
        if constexpr (I < NVI) {
            constexpr SItem it = sval_item<L>(I);
            const f32x4* const pkt = park + lane_t;
            static_for<(it.l3 == 0 ? 1 : it.na)>([&]<int a>() {      // (scalar outputs: one B operand for both tiles)
                if constexpr (park_packed<L>(it.bq[a])) {      // one slot: {hi(0,1), hi(4,5), lo(0,1), lo(4,5)}, the other registers are the zero padding of 8x3e
                    const f32x4 sp = pkt[(park_phys<L>(it.bq[a])) * 64];
                    o.h[a] = f32x4{sp[0], 0.0f, sp[1], 0.0f};
                    if constexpr (!HP) o.l[a] = f32x4{sp[2], 0.0f, sp[3], 0.0f};
                } else {
                o.h[a] = pkt[(park_phys<L>(it.bq[a])) * 64];
                if constexpr (!HP) o.l[a] = pkt[(park_phys<L>(it.bq[a]) + 1) * 64];
                }
                if constexpr (it.neg[a]) {
                    o.h[a] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, o.h[a]) ^ 0x80008000u);
                    if constexpr (!HP) o.l[a] = __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, o.l[a]) ^ 0x80008000u);
                }
            });
        }
        return o;
    };
    f32x16 V0[2][2], V1[2][3], V2[2][5], V3[2][4];      // (paired tiles, lmax 3: V2 uses three, V3 four)
    // ONE: every edge of the tile has cut-off factor 1 (no edge shorter than r_mincut_nonscalar_sh -- all but a handful of tiles): set 1 accumulates
    // straight into set 0's tiles, and a completed degree is read once instead of twice (-240 VALU instructions per tile)
    auto run_item = [&]<int I, bool ONE>(const SB& b) {
        constexpr SItem it = sval_item<L>(I);
        constexpr int ST = ONE ? 0 : it.set;
        auto& V = [&]() -> auto& { if constexpr (it.l3 == 0) return V0; else if constexpr (it.l3 == 1) return V1; else if constexpr (it.l3 == 2) return V2; else return V3; }();
        static_for<it.na>([&]<int a>() {
            constexpr bool fst = ONE ? it.first_one[a] : it.first[a];
            f32x16 init = {};
            if constexpr (fst && it.l3 == 0 && it.set == 0) init = ldrows_lds(rows, hi, RL::val0, it.acc[a]);      // sep_value.lin's bias
            V[ST][it.tile[a]] = mfma_h(__builtin_bit_cast(h8, it.up[a] ? aring[it.aslot[a] % RS].hu : aring[it.aslot[a] % RS].h), __builtin_bit_cast(h8, b.h[it.l3 == 0 ? 0 : a]), fst ? init : V[ST][it.tile[a]]);
        });
        if constexpr (!HP) {
            static_for<it.na>([&]<int a>() { V[ST][it.tile[a]] = mfma_h(__builtin_bit_cast(h8, it.up[a] ? aring[it.aslot[a] % RS].hu : aring[it.aslot[a] % RS].h), __builtin_bit_cast(h8, b.l[it.l3 == 0 ? 0 : a]), V[ST][it.tile[a]]); });
            static_for<it.na>([&]<int a>() { V[ST][it.tile[a]] = mfma_h(__builtin_bit_cast(h8, it.up[a] ? aring[it.aslot[a] % RS].lu : aring[it.aslot[a] % RS].l), __builtin_bit_cast(h8, b.h[it.l3 == 0 ? 0 : a]), V[ST][it.tile[a]]); });
        }
    };
    // a completed degree: set0 + cns . set1, back to the global frame, into val<l3> (what store_group reads)
    auto finish_value = [&]<int l3, bool ONE>() {
        if constexpr (l3 == 0) {
            static_for<2>([&]<int T>() { static_for<16>([&]<int R>() {
                float v = (ONE || sval_acc_used<L>(0, 0, T)) ? V0[0][T][R] : 0.0f;
                if constexpr (!ONE && sval_acc_used<L>(0, 1, T)) v += cns * V0[1][T][R];
                val0[T][R] = v;
            }); });
        } else {
            constexpr int d3 = 2 * l3 + 1, NR = mul_of(l3) >= 32 ? 16 : mul_of(l3) / 2;
            constexpr bool PR = acc_paired<L>(l3);      // component K in rows 16 (K % 2) .. of tile K / 2
            auto& V = [&]() -> auto& { if constexpr (l3 == 1) return V1; else if constexpr (l3 == 2) return V2; else return V3; }();
            static_for<NR>([&]<int R>() {
                if constexpr (pad_reg<L, NW>(l3, R)) static_for<d3>([&]<int K>() { if constexpr (l3 == 3) val3[K][R] = 0.0f; else if constexpr (l3 == 2) val2[K][R] = 0.0f; else val1[K][R] = 0.0f; });
                else {
                float v[d3];
                static_for<d3>([&]<int K>() {
                    constexpr int T = PR ? K / 2 : K, Rr = PR ? 8 * (K % 2) + R : R;
                    float x = 0.0f;
                    if constexpr (ONE) x = V[0][T][Rr];
                    else {
                    if constexpr (sval_acc_used<L>(l3, 0, K)) x = V[0][T][Rr];
                    if constexpr (sval_acc_used<L>(l3, 1, K)) x += cns * V[1][T][Rr];
                    }
                    v[K] = x;
                });
                Rot<l3>::out(v, tg);
                static_for<d3>([&]<int K>() { if constexpr (l3 == 1) val1[K][R] = v[K]; else if constexpr (l3 == 2) val2[K][R] = v[K]; else val3[K][R] = v[K]; });
                if constexpr (R == NR - 4 + pad_live<L, NW>(l3) - 1) tok = v[0];
                }
            });
        }
    };
    auto value_loop = [&]<bool ONE>() {
    static_for<PDV>([&]<int I>() { load_A_of.template operator()<I>(); });
    SB vb_cur = load_B.template operator()<0>();
    // A completed degree leaves in pieces, one per pipeline region, under the GEMMs of the degree that follows it in the walk (highest degree first,
    // the scalars last): piece 0 = read the accumulators, set0 + cns . set1, rotate back (finish_value); piece q >= 1 = the segmented reduction and
    // the record stores of component (output tile) q - 1 (store_group).  Region R carries item R and every piece that is due: piece q of degree g
    // in region sval_group_last(g) + 1 + q.  Only the scalars' pieces are left without GEMMs beside them (the tail).
    constexpr auto npieces = [](int l3) { return 1 + (l3 == 0 ? 2 : 2 * l3 + 1); };
    constexpr int NTAIL = npieces(0);
#ifndef DEDF_SGBV_SO2
#define DEDF_SGBV_SO2 0
#endif
    static_for<NVI + NTAIL>([&]<int I>() {
        // (operand requests anchored to the stream: dedf_dev.h::tie on a word of the B operand this region's MFMAs read)
        if constexpr (I < NVI) tok = __builtin_bit_cast(f32x4, vb_cur.h[0])[0];
        if constexpr (I + PDV < NVI) { lane16_t = tie(wv.lane16, tok); lane16_r16_t = tie(wv.lane16_r16, tok); if constexpr (L == 3) lane16_r16up_t = tie(wv.lane16_r16up, tok); }
        if constexpr (I + 1 < NVI) lane_t = tie(wv.lane, tok);
        load_A_of.template operator()<I + PDV>();
        const SB b_nxt = load_B.template operator()<I + 1>();
        sched_fence();
        if constexpr (I < NVI) run_item.template operator()<I, ONE>(vb_cur);
        static_for<L + 1>([&]<int g>() {
            constexpr int q = I - 1 - sval_group_last<L>(g);
            if constexpr (q >= 0 && q < npieces(g)) {
                if constexpr (q == 0) finish_value.template operator()<g, ONE>();
                else store_group.template operator()<g, q - 1, 1>();
            }
        });
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (DEDF_SGBV_SO2 > 0 && I < NVI) static_for<16>([&]<int i>() { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, DEDF_SGBV_SO2, 0); });
#endif
        sched_fence();
        vb_cur = b_nxt;
        if constexpr (I == sval_group_last<L>(L) + 1) DEDF_STAMP(10);
        if constexpr (L >= 2 && I == sval_group_last<L>(1) + 1) DEDF_STAMP(13);
    });
    };
#ifndef DEDF_VAL_ONESET
#define DEDF_VAL_ONESET 1
#endif
    if (DEDF_VAL_ONESET && __all(!valid || cns == 1.0f)) value_loop.template operator()<true>();
    else value_loop.template operator()<false>();
    } else {
    // Software pipeline over the work items: region I requests the A slots of item I + 2 and the parked B chunks of item I + 1,
    // runs the MFMAs of item I and, beside them, the contraction of the accumulators item I - 1 completed.
#ifndef DEDF_V_PDA
#define DEDF_V_PDA 2
#endif
#ifndef DEDF_V_LAG
#define DEDF_V_LAG 1
#endif
    constexpr int NVI = val_num_items<L>(), PDV = (L == 2 && MODE == 1 && F0 == 128) ? DEDF_V_PDA_TAB : DEDF_V_PDA, RS = 2 * (PDV + 1), LAG = DEDF_V_LAG;
    const int o_S_val = opaque_s(P.o_S_val);
    struct ASlot { f32x4 h, l; };
    struct BSet { f32x4 h[3], l[3]; };
    ASlot aring[RS];
    // Operand requests are anchored to the VALU stream (dedf_dev.h::tie) through `tok`, a value the previous region's contraction
    // produced: hipcc otherwise lets the MFMAs / VALU work sink below the fences while the requests stay, and a dozen operand sets
    // end up in flight (and spilled).
    float tok = logit0;
    int lane16_t = wv.lane16, lane16_r16_t = wv.lane16_r16, lane_t = wv.lane;
    auto load_A = [&]<int S>() {
        if constexpr (S < val_num_slots<L>()) {
            const int lv = mul_of(val_slot_l3<L>(S)) < 32 ? lane16_r16_t : lane16_t;      // rows 16-31 are padding: half the lanes fetch
            aring[S % RS].h = bldw(wv, lv, (o_S_val + S * 512) * 4);
            if constexpr (!HP) aring[S % RS].l = bldw(wv, lv, (o_S_val + S * 512 + 256) * 4);
        }
    };
    auto load_A_of = [&]<int I>() {
        if constexpr (I < NVI) static_for<val_item<L>(I).new_slots>([&]<int n>() { load_A.template operator()<val_item_slot0<L>(I) + n>(); });
    };
    // INPUT-side items (dedf_net.h: the chained l3 = 0 group at every lmax; at lmax 3 the paths l1 = 3 -> l3 <= 2): B_k = the Clebsch-Gordan contraction
    // of the parked components of a K-chunk with the SH, split like every other B operand.  The components come back from their parked halves
    // (hi + lo, one v_fma_mix each); those of the packed 8x3e block (registers 0, 1, 4, 5 of the chunk) are kept over a run of such items.
    float uin[7][4];
    auto load_B = [&]<int I>() {
        BSet o{};
        if constexpr (I < NVI && val_item<L>(I < NVI ? I : 0).in_side) {
            constexpr VItem it = val_item<L>(I);
            constexpr PathInfo pi = dtp_path<L>(it.p);
            constexpr int l1 = pi.l1, d1 = 2 * l1 + 1, d3 = 2 * pi.l3 + 1;
            using Cg = CG<pi.l1, pi.l2, pi.l3>;
            float m[Cg::NM];
            Cg::make(Y.template get<pi.l2>(), m);
            const f32x4* const pkt = park + lane_t;
            if constexpr (park_packed<L>(it.bq[0])) {
                static_assert(L == 3 && l1 == 3);
                constexpr VItem prev = val_item<L>(I - 1);
                if constexpr (!(prev.in_side && dtp_path<L>(prev.p >= 0 ? prev.p : 0).l1 == 3 && I >= 1)) {
                    static_for<7>([&]<int i>() {
                        const u32x4 s = __builtin_bit_cast(u32x4, pkt[(park_phys<L>(park_slot<L>(3, i, 0))) * 64]);      // {hi(0,1), hi(4,5), lo(0,1), lo(4,5)}
                        static_for<4>([&]<int c>() { uin[i][c] = unsplit<c % 2>(s[c / 2], s[2 + c / 2]); });
                    });
                }
                float t[4][d3];
                static_for<4>([&]<int c>() {
                    if constexpr (pad_reg<L, NW>(3, c & 1)) static_for<d3>([&]<int K>() { t[c][K] = 0.0f; });      // (narrow UNet level: one true channel per pair)
                    else {
                        const float x[7] = {uin[0][c], uin[1][c], uin[2][c], uin[3][c], uin[4][c], uin[5][c], uin[6][c]};
                        Cg::apply(x, m, t[c]);
                    }
                });
                static_for<it.share_b ? 1 : it.na>([&]<int a>() {
                    constexpr int K = it.comp[a];
                    const float v4[4] = {t[0][K], t[1][K], t[2][K], t[3][K]};
                    const f32x4 s = split4pk(v4);
                    o.h[a] = f32x4{s[0], 0.0f, s[1], 0.0f};
                    if constexpr (!HP) o.l[a] = f32x4{s[2], 0.0f, s[3], 0.0f};
                });
            } else {      // a 16-channel chunk of degree 1 or 2 (chained items only): hi halves in one slot, residuals in the next
                static_assert(it.chain && l1 >= 1 && l1 <= 2);
                constexpr int KCH = mul_of(l1) / 16;      // park_slot(l, i, c) = park_slot(l, 0, c) + i * KCH
                float u[d1][8];
                static_for<d1>([&]<int i>() {
                    const h8 hh = __builtin_bit_cast(h8, pkt[(park_phys<L>(it.bq[0] + i * KCH)) * 64]);
                    if constexpr (HP) static_for<8>([&]<int r>() { u[i][r] = (float)hh[r]; });
                    else {
                        const u32x4 hw = __builtin_bit_cast(u32x4, hh), lw = __builtin_bit_cast(u32x4, pkt[(park_phys<L>(it.bq[0] + i * KCH) + 1) * 64]);
                        static_for<8>([&]<int r>() { if constexpr (!pad_reg<L, NW>(l1, r)) u[i][r] = unsplit<r % 2>(hw[r / 2], lw[r / 2]); else u[i][r] = 0.0f; });
                    }
                });
                float v[d3][8];
                static_for<8>([&]<int r>() {
                    if constexpr (pad_reg<L, NW>(l1, r)) static_for<d3>([&]<int K>() { v[K][r] = 0.0f; });
                    else {
                        float x[d1], t[d3];
                        static_for<d1>([&]<int i>() { x[i] = u[i][r]; });
                        Cg::apply(x, m, t);
                        static_for<d3>([&]<int K>() { v[K][r] = t[K]; });
                    }
                });
                static_for<it.share_b ? 1 : it.na>([&]<int a>() {
                    HL sp;
                    if constexpr (pad_reg<L, NW>(l1, 2)) sp = split8zx<HP>(v[it.comp[a]]); else sp = split8x<HP>(v[it.comp[a]]);
                    o.h[a] = __builtin_bit_cast(f32x4, sp.hi);
                    if constexpr (!HP) o.l[a] = __builtin_bit_cast(f32x4, sp.lo);
                });
            }
        } else if constexpr (I < NVI) {
            constexpr VItem it = val_item<L>(I);
            const f32x4* const pkt = park + lane_t;
            static_for<it.share_b ? 1 : it.na>([&]<int a>() {
                if constexpr (park_packed<L>(it.bq[a])) {      // one slot: {hi(0,1), hi(4,5), lo(0,1), lo(4,5)}, the other registers are zeros
                    const f32x4 s = pkt[(park_phys<L>(it.bq[a])) * 64];
                    o.h[a] = f32x4{s[0], 0.0f, s[1], 0.0f};
                    if constexpr (!HP) o.l[a] = f32x4{s[2], 0.0f, s[3], 0.0f};
                } else {
                    o.h[a] = pkt[(park_phys<L>(it.bq[a])) * 64];
                    if constexpr (!HP) o.l[a] = pkt[(park_phys<L>(it.bq[a]) + 1) * 64];
                }
            });
        }
        return o;
    };
    auto run_item = [&]<int I>(const BSet& b, f32x16 (&G)[3]) {
        constexpr VItem it = val_item<L>(I);
        const f32x16 zero = {};
        static_for<it.na>([&]<int a>() {      // (chained items: one B operand, accumulator a = output tile a)
            G[a] = mfma_h(__builtin_bit_cast(h8, aring[it.aslot[a] % RS].h), __builtin_bit_cast(h8, b.h[it.share_b ? 0 : a]), it.first ? zero : G[a]);
        });
        if constexpr (!HP) {
            static_for<it.na>([&]<int a>() { G[a] = mfma_h(__builtin_bit_cast(h8, aring[it.aslot[a] % RS].h), __builtin_bit_cast(h8, b.l[it.share_b ? 0 : a]), G[a]); });
            static_for<it.na>([&]<int a>() { G[a] = mfma_h(__builtin_bit_cast(h8, aring[it.aslot[a] % RS].l), __builtin_bit_cast(h8, b.h[it.share_b ? 0 : a]), G[a]); });
        }
    };
    auto contract = [&]<int I>(f32x16 (&G)[3]) {      // value[.., k] += (sum_j C_ijk Y_j) G_i   for the components this item completed
        constexpr VItem it = val_item<L>(I);
        constexpr PathInfo pi = dtp_path<L>(it.p);
        constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d3 = 2 * l3 + 1;
        constexpr int NR = mul_of(l3) >= 32 ? 16 : mul_of(l3) / 2;       // registers holding valid rows
        using Cg = CG<l1, l2, l3>;
        float m[Cg::NM];
        Cg::make(Y.template get<l2>(), m);
        constexpr int R_LAST = NR - 4 + pad_live<L, NW>(l3) - 1;      // last register that holds a true channel
        if constexpr (it.in_side || it.chain) {      // the accumulators ARE output components: add them
            static_assert((l3 == 0 || !val_item_opens_group<L>(I)) && l3 <= 2);
            static_for<NR>([&]<int R>() {
                if constexpr (!pad_reg<L, NW>(l3, R)) static_for<it.na>([&]<int a>() {
                    constexpr int K = it.comp[a];
                    float o;
                    if constexpr (l3 == 0) o = val0[it.share_b ? a : it.t][R] + G[a][R]; else if constexpr (l3 == 1) o = val1[K][R] + G[a][R]; else o = val2[K][R] + G[a][R];
                    opaque_v(o);
                    if constexpr (l3 == 0) val0[it.share_b ? a : it.t][R] = o; else if constexpr (l3 == 1) val1[K][R] = o; else val2[K][R] = o;
                    if constexpr (R == R_LAST && a == 0) tok = o;
                });
            });
        } else
        static_for<NR>([&]<int R>() {
            if constexpr (pad_reg<L, NW>(l3, R)) {      // a zero-padding output channel: nothing to contract (l3 = 0: the accumulator keeps its zero bias)
                if constexpr (val_item_opens_group<L>(I)) static_for<d3>([&]<int K>() {
                    if constexpr (l3 == 1) val1[K][R] = 0.0f; else if constexpr (l3 == 2) val2[K][R] = 0.0f; else if constexpr (l3 == 3) val3[K][R] = 0.0f;
                });
            } else {
            float o[d3];
            static_for<d3>([&]<int K>() {
                if constexpr (l3 == 0) o[K] = val0[it.t][R];
                else if constexpr (val_item_opens_group<L>(I)) o[K] = 0.0f;
                else if constexpr (l3 == 1) o[K] = val1[K][R];
                else if constexpr (l3 == 2) o[K] = val2[K][R];
                else o[K] = val3[K][R];
            });
            if constexpr (it.merge) Cg::template acc<0>(G[0][R] + G[1][R], m, o);
            else static_for<it.na>([&]<int a>() { Cg::template acc<it.comp[a]>(G[a][R], m, o); });
            // (opaque: the accumulation happens HERE -- hipcc otherwise defers it to the store of the block and keeps every G tile alive)
            static_for<d3>([&]<int K>() {
                opaque_v(o[K]);
                if constexpr (l3 == 0) val0[it.t][R] = o[K]; else if constexpr (l3 == 1) val1[K][R] = o[K]; else if constexpr (l3 == 2) val2[K][R] = o[K]; else val3[K][R] = o[K];
            });
            if constexpr (R == R_LAST) tok = o[0];
            }
        });
    };
    static_for<PDV>([&]<int I>() { load_A_of.template operator()<I>(); });
    BSet vb_cur = load_B.template operator()<0>();
    // G: the accumulator tiles of the item in flight; Gq: the tiles of completed items waiting for their contraction, LAG regions behind the MFMAs that
    // finished them (a ring of LAG sets: the contraction of region I reads what item I - LAG left, so with LAG >= 2 no VALU instruction of a region
    // waits for an MFMA of the region before)
    f32x16 G[3], Gq[LAG > 0 ? LAG : 1][3];
    static_for<NVI + LAG>([&]<int I>() {
        lane16_t = tie(wv.lane16, tok); lane16_r16_t = tie(wv.lane16_r16, tok); lane_t = tie(wv.lane, tok);
        load_A_of.template operator()<I + PDV>();
        // (input-side operands are FORMED, not just fetched: that VALU work belongs into the region of this item's MFMAs, behind the fence)
#ifndef DEDF_VAL_LATE_B
#define DEDF_VAL_LATE_B 1
#endif
        // (the sampler's timed instantiations only: -0.5 % / -1.6 % at lmax 2 / 3; the lmax-3 UNet layer kernel pays 32 B more scratch for it: 10.5 -> 11-12.7 ms per forward)
        constexpr bool late_b = DEDF_VAL_LATE_B && MODE == 1 && I + 1 < NVI && val_item<L>(I + 1 < NVI ? I + 1 : 0).in_side;
        BSet b_nxt{};
        if constexpr (!late_b) b_nxt = load_B.template operator()<I + 1>();
        sched_fence();
        if constexpr (late_b) b_nxt = load_B.template operator()<I + 1>();
        constexpr int F = I - LAG;                       // item whose accumulators are contracted in this region
        constexpr bool fin = F >= 0 && val_item<L>(F).last;
        if constexpr (LAG > 0 && I >= 1 && I - 1 < NVI) { if constexpr (val_item<L>(I - 1).last) static_for<3>([&]<int a>() { Gq[(I - 1) % (LAG > 0 ? LAG : 1)][a] = G[a]; }); }
        if constexpr (I < NVI) run_item.template operator()<I>(vb_cur, G);
        if constexpr (fin) {
            if constexpr (LAG > 0) contract.template operator()<F>(Gq[F % (LAG > 0 ? LAG : 1)]); else contract.template operator()<F>(G);
            constexpr int ge = val_item<L>(F).group_end;
            if constexpr (ge >= 0 && ge < L) store_group.template operator()<ge>();
        }
        sched_fence();
        vb_cur = b_nxt;
        if constexpr (F >= 0 && val_item<L>(F).group_end == 0) DEDF_STAMP(10);
        if constexpr (L >= 2 && F >= 0 && val_item<L>(F).group_end == 1) DEDF_STAMP(13);
        if constexpr (L == 3 && F >= 0 && val_item<L>(F).group_end == 2) DEDF_STAMP(2);      // (slot 2 is free in the table-reading kernel)
    });
    }
    DEDF_STAMP(14);
    if constexpr (!SO2) store_group.template operator()<L>();
    if constexpr (DBG) if (P.dbg_out != nullptr && valid && hi == 0) st4(drec_of() + D, f32x4{logit[0], logit[1], logit[2], logit[3]});
    if constexpr (MODE == 1) {
        geo.ok = e_next >= 0; geo.src = nsrc; geo.dst = ndst;
        geo.vx = nk[0] - nq[0]; geo.vy = nk[1] - nq[1]; geo.vz = nk[2] - nq[2];
    }
    DEDF_STAMP(11);
}

}  // namespace dedf
