// Fused per-edge pipeline of the equivariant graph attention (one wave = one tile of 32 edges of one scale).
//
// Restates, per edge, reference graph_parser.py:146-224 (geometry, soft cut-offs, length embedding, spherical
// harmonics), multiscale_tensor_field.py:225-234 (edge pre-linear with the time embedding), and
// graph_attention.py:231-247 (radial MLP -> depth-wise TP -> {sep_alpha | lin -> Gate -> depth-wise TP -> lin}).
// Nothing per-edge except the result record (value in internal layout + one logit per head, 244 floats at lmax 2)
// reaches HBM: the 480 radial weights, the two 1568-float TP outputs and all activations stay in registers.
//
// Layout: lane = (edge column = lane & 31, row half hi = lane >> 5); see dedf_layout.h.  Every dense layer is
// D[out][edge] += W[out][k] * act[k][edge] on v_mfma_f32_32x32x2_f32 with the previous layer's accumulator
// registers used directly as B operands; weights stream from L2 as pre-permuted float4 per lane through one
// buffer descriptor.
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
    int o_A_lin[4];           // l3 = 0: lin0 rows + alpha rows; l3 >= 1: mul(l3) rows
    int o_b_r0;               // row-packed bias over the l3 = 0 row space
    int o_A_val[4];           // sep_value.lin with the shared DTP weights folded in
    int o_b_val0;             // row-packed (64)
    int o_alpha_dot;          // row-packed over the two alpha tiles
    float* out;               // [E][edge_rec]
    float* dbg_w;             // optional [E][WN] dump of the radial weights (tests)
    unsigned long long* phase_prof;   // optional [grid][16] per-wave phase cycle sums (built with -DDEDF_PHASE_PROF)
};

template <int L> struct SH {           // spherical harmonics of one edge, non-scalar blocks already cut off
    float y0[1], y1[3], y2[5], y3[7];
    template <int l> DEDF_DEV const float* get() const {
        if constexpr (l == 0) return y0; else if constexpr (l == 1) return y1;
        else if constexpr (l == 2) return y2; else return y3;
    }
};

// LayerNorm over NT*32 channels of one item (rows split over lane and lane^32) followed by SiLU
template <int NT>
DEDF_DEV void ln_silu(f32x16 (&x)[NT], const Wave& wv, int o_gamma, int o_beta) {
    float s = 0.0f;
    static_for<NT>([&]<int T>() { static_for<16>([&]<int R>() { s += x[T][R]; }); });
    s += xor32(s);
    const float mean = s * (1.0f / (NT * 32));
    float v = 0.0f;
    static_for<NT>([&]<int T>() { static_for<16>([&]<int R>() { const float d = x[T][R] - mean; v += d * d; }); });
    v += xor32(v);
    const float rstd = 1.0f / sqrtf(v * (1.0f / (NT * 32)) + 1e-5f);
    static_for<NT>([&]<int T>() {
        const f32x16 g = ldrows(wv, o_gamma, T), b = ldrows(wv, o_beta, T);
        static_for<16>([&]<int R>() { x[T][R] = siluf((x[T][R] - mean) * rstd * g[R] + b[R]); });
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
#define DEDF_PROF_ARG , unsigned long long (&pacc)[12]
#else
#define DEDF_PROF_ARG
#endif

template <int L, int F0>
DEDF_DEV void edge_tile(const EdgeParams& P, const Wave& wv, int scale, int e0, int n_valid DEDF_PROF_ARG) {
#if defined(DEDF_PHASE_PROF) && defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t_last = __builtin_readcyclecounter();
#endif
    constexpr int D = feat_dim<L>();
    constexpr int REC = edge_rec<L>();
    constexpr int WN = dtp_wn<L>();
    constexpr int NWT = WN / 32;
    constexpr int NR0 = r0_tiles<L>();
    const int hi = wv.hi;
    // weight-image offsets, re-materialised per tile (see opaque_s)
    const int o_A_r1 = opaque_s(P.o_A_r1), o_b_r1 = opaque_s(P.o_b_r1), o_g_r1 = opaque_s(P.o_g_r1), o_be_r1 = opaque_s(P.o_be_r1);
    const int o_A_r2 = opaque_s(P.o_A_r2), o_b_r2 = opaque_s(P.o_b_r2), o_g_r2 = opaque_s(P.o_g_r2), o_be_r2 = opaque_s(P.o_be_r2);
    const int o_A_r3 = opaque_s(P.o_A_r3), o_off_r3 = opaque_s(P.o_off_r3), o_b_r0 = opaque_s(P.o_b_r0);
    const int o_A_r1_l = opaque_s(P.o_A_r1_l), o_A_r2_l = opaque_s(P.o_A_r2_l), o_A_r3_l = opaque_s(P.o_A_r3_l);
    const int o_b_val0 = opaque_s(P.o_b_val0), o_alpha_dot = opaque_s(P.o_alpha_dot);
    int o_A_lin[L + 1], o_A_val[L + 1];
    static_for<L + 1>([&]<int l>() { o_A_lin[l] = opaque_s(P.o_A_lin[l]); o_A_val[l] = opaque_s(P.o_A_val[l]); });
    const bool valid = wv.col < n_valid;
    const int e = e0 + (valid ? wv.col : 0);
    const int src = P.edge_src[e], dst = P.edge_dst[e];
    const int pose = dst / P.nQ;

    // ---- geometry (graph_parser.py:159-215) ---------------------------------------------------------------------
    const float vx = P.key_x[3 * src + 0] - P.qpos[3 * dst + 0];
    const float vy = P.key_x[3 * src + 1] - P.qpos[3 * dst + 1];
    const float vz = P.key_x[3 * src + 2] - P.qpos[3 * dst + 2];
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);
    const float radius = P.radius[scale];
    float logit0 = 0.0f;
    if (radius > 0.0f) {
        const float cut = 1.0f - soft_step((len - P.cut_begin[scale]) / P.cut_div[scale]);
        logit0 = logf(fmaxf(cut, 1e-12f));
    }
    const float cns = soft_step((len - P.ns_lo) / P.ns_div);
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
        static_assert(L <= 2, "l = 3 spherical harmonics are a next-row item");
    }

    // ---- length embedding: this lane's 32 of the 64 channels (k = s + 32*hi) ------------------------------------------
    float eb[32];
    {
        const int o_enc = P.o_enc + scale * 192;
        const int hi128 = hi * 128;
        if (radius > 0.0f) {           // GaussianRadialBasis, radial_func.py:208-227
            const float t = len / radius;
            static_for<8>([&]<int G>() {
                const f32x4 mu = bld4(wv.w, hi128, (o_enc + 4 * G) * 4);
                const f32x4 is = bld4(wv.w, hi128, (o_enc + 64 + 4 * G) * 4);
                const f32x4 w = bld4(wv.w, hi128, (o_enc + 128 + 4 * G) * 4);
                static_for<4>([&]<int J>() {
                    const float z = (t - mu[J]) * is[J];
                    eb[4 * G + J] = fexp(-0.5f * (z * z)) * w[J];
                });
            });
        } else {                       // SinusoidalPositionEmbeddings(n = 1000), radial_func.py:305-316
            const float x = len / P.len_enc_max_r * 1000.0f;
            static_for<8>([&]<int G>() {
                const f32x4 fr = bld4(wv.w, 0, (o_enc + 4 * G) * 4);
                static_for<4>([&]<int J>() { eb[4 * G + J] = sin_or_cos(x * fr[J], hi); });
            });
        }
    }

    DEDF_STAMP(0);
    // ---- edge pre-linear + SiLU (multiscale_tensor_field.py:225-234); time part + bias arrive as per-pose rows --------
    constexpr int NH = F0 / 32;      // pre-linear width: 128 (length + time embedding) or 64 (EBM critic: length only)
    f32x16 h[NH];
    {
        const Buf tbb = make_buf(P.tb, P.tb_bytes);
        const int tvoff = (pose * P.tb_pose_stride) * 4 + wv.hi64;
        const int oA = opaque_s(P.o_A_pre + scale * (NH * 4 * 256)), oAl = opaque_s(P.o_A_pre_l + scale * (NH * 4 * 256));
        static_for<NH>([&]<int To>() { h[To] = ldrows(tbb, tvoff, scale * F0, To); });
        dense_rot_h<NH, 4, 2>(wv, oA, oAl, h, [&]<int c, int j>() { return eb[8 * c + j]; });
        static_for<NH>([&]<int To>() { static_for<16>([&]<int R>() { h[To][R] = siluf(h[To][R]); }); });
    }
    DEDF_STAMP(1);
    // ---- RadialProfile layers 1, 2 (equiformer/radial_func.py:11-60) ---------------------------------------------------
    f32x16 r1[4];
    static_for<4>([&]<int To>() { r1[To] = ldrows(wv, o_b_r1, To); });
    dense_rot_h<4, F0 / 16, 2>(wv, o_A_r1, o_A_r1_l, r1, [&]<int c, int j>() { return h[c / 2][8 * (c % 2) + j]; });
    DEDF_STAMP(2);
    ln_silu<4>(r1, wv, o_g_r1, o_be_r1);
    DEDF_STAMP(3);
    f32x16 r2[2];
    static_for<2>([&]<int To>() { r2[To] = ldrows(wv, o_b_r2, To); });
    dense_rot_h<2, 8, 2>(wv, o_A_r2, o_A_r2_l, r2, [&]<int c, int j>() { return r1[c / 2][8 * (c % 2) + j]; });
    DEDF_STAMP(4);
    ln_silu<2>(r2, wv, o_g_r2, o_be_r2);
    DEDF_STAMP(5);

    // ---- layer 3 (-> per-edge TP weights, one 32-row tile at a time) fused with DTP #1 and the lin / sep_alpha GEMMs ----
    // accumulators: l3 = 0 -> NR0 tiles (lin scalars+gates | alpha), l3 >= 1 -> one tile per m
    // the 16-channel l3 = 2 outputs run on 16x16x4 MFMAs: [m][edge sub-tile 0-15 | 16-31], 4 registers each
    constexpr int NACC = NR0 + (L >= 1 ? 3 : 0);
    constexpr int AB1 = NR0;                  // first accumulator tile of the l3 = 1 outputs
    f32x16 acc[NACC];
    f32x4 acc2[5][2];
    static_for<5>([&]<int K>() { static_for<2>([&]<int S>() { acc2[K][S] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }); });
    static_for<NR0>([&]<int T>() { acc[T] = ldrows(wv, o_b_r0, T); });
    static_for<NACC - NR0>([&]<int T>() { static_for<16>([&]<int R>() { acc[NR0 + T][R] = 0.0f; }); });
    const Buf msgb = make_buf(P.msg, P.msg_bytes);
    // per-l1 lane offsets of the 4 message rows this lane owns inside an 8-row group
    const int mv0 = src * (D * 4) + hi * 16, mv1 = src * (D * 4) + hi * 48, mv2 = src * (D * 4) + hi * 80;

    // Software pipeline over the WN/8 depth-wise-TP groups (8 weight rows = 4 K-steps each).  Region G issues, in one
    // scheduling region so that hipcc interleaves them:  loads for G+1 / G+2,  the layer-3 MFMAs of the NEXT weight tile,
    // the lane-local Clebsch-Gordan VALU work of group G+1 (-> its B operands),  and the lin / sep_alpha MFMAs of group G.
    constexpr int NGRP = WN / 8;
    struct AOps { f32x4 a[NR0]; };
    struct XOps { f32x4 x[2 * L + 1]; };
    struct BOps { float b[4][2 * L + 1]; };
    auto load_A = [&]<int G>() {
        AOps o;
        if constexpr (G < NGRP) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l3 = pi.l3, gi = dtp_group_index<L>(G), NG = dtp_k<L>(l3) / 8;
            if constexpr (l3 == 0) static_for<NR0>([&]<int To>() { o.a[To] = lda(wv, o_A_lin[0], NG, To, gi); });
            else if constexpr (l3 == 2) { const f32x2 t = lda16(wv, o_A_lin[2], gi); o.a[0][0] = t[0]; o.a[0][1] = t[1]; }
            else o.a[0] = lda(wv, o_A_lin[l3], NG, 0, gi);
        }
        return o;
    };
    auto load_X = [&]<int G>() {      // this lane's 4 source-message rows of the group (contiguous in the reference layout)
        XOps o;
        if constexpr (G < NGRP) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l1 = pi.l1, d1 = 2 * l1 + 1, u0 = G * 8 - pi.wstart;
            const int mv = l1 == 0 ? mv0 : (l1 == 1 ? mv1 : mv2);
            static_for<d1>([&]<int Q>() { o.x[Q] = bld4(msgb, mv, (blk_off(l1) + u0 * d1 + 4 * Q) * 4); });
        }
        return o;
    };
    auto valu_group = [&]<int G>(const XOps& xo, const f32x16& wtile) {
        BOps o;
        if constexpr (G < NGRP) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d1 = 2 * l1 + 1, d3 = 2 * l3 + 1, g = G % 4;
            using C = CG<l1, l2, l3>;
            float m[C::NM];
            C::make(Y.template get<l2>(), m);
            float xr[4 * d1];
            static_for<d1>([&]<int Q>() {
                xr[4 * Q] = xo.x[Q][0]; xr[4 * Q + 1] = xo.x[Q][1]; xr[4 * Q + 2] = xo.x[Q][2]; xr[4 * Q + 3] = xo.x[Q][3];
            });
            static_for<4>([&]<int j>() {
                float t[d3];
                C::apply(&xr[j * d1], m, t);
                static_for<d3>([&]<int K>() { o.b[j][K] = t[K] * wtile[4 * g + j]; });
            });
        }
        return o;
    };
    auto mfma_dtp = [&]<int G>(const AOps& ao, const BOps& bo) {
        constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
        constexpr int l3 = pi.l3, d3 = 2 * l3 + 1;
        if constexpr (l3 == 0) {
            static_for<NR0>([&]<int To>() { mfma_group(acc[To], ao.a[To], bo.b[0][0], bo.b[1][0], bo.b[2][0], bo.b[3][0]); });
        } else if constexpr (l3 == 1) {
            static_for<d3>([&]<int K>() { mfma_group(acc[AB1 + K], ao.a[0], bo.b[0][K], bo.b[1][K], bo.b[2][K], bo.b[3][K]); });
        } else {
            static_for<2>([&]<int pr>() {
                float x[5], y[5];
                static_for<5>([&]<int K>() { x[K] = bo.b[2 * pr][K]; y[K] = bo.b[2 * pr + 1][K]; });
                swap16x5(x, y);                // x: edges 0-15, y: edges 16-31, K-slots in 16-lane rows
                static_for<5>([&]<int K>() {
                    acc2[K][0] = mfma16(ao.a[0][pr], x[K], acc2[K][0]);
                    acc2[K][1] = mfma16(ao.a[0][pr], y[K], acc2[K][1]);
                });
            });
        }
    };
    // layer 3 on split-fp16 MFMAs: r2 (64 rows = 4 chunks) is split once per edge tile; per weight tile 4 chunks x 3 MFMAs.
    // A operands (hi and lo image) form one global stream over all tiles, PD3 chunks ahead.
    HL r2s[4];
    static_for<4>([&]<int c>() {
        float t[8];
        static_for<8>([&]<int J>() { t[J] = r2[c / 2][8 * (c % 2) + J]; });
        r2s[c] = split8(t);
    });
    constexpr int NL3 = NWT * 4, PD3 = 2;
    f32x4 l3h[PD3], l3l[PD3];
    static_for<PD3>([&]<int I>() { l3h[I] = bld4(wv.w, wv.lane16, (o_A_r3 + I * 256) * 4); l3l[I] = bld4(wv.w, wv.lane16, (o_A_r3_l + I * 256) * 4); });
    auto l3_chunk = [&]<int I>(f32x16& w) {
        constexpr int c = I % 4;
        const h8 ah = __builtin_bit_cast(h8, l3h[I % PD3]), al = __builtin_bit_cast(h8, l3l[I % PD3]);
        if constexpr (I + PD3 < NL3) {
            l3h[I % PD3] = bld4(wv.w, wv.lane16, (o_A_r3 + (I + PD3) * 256) * 4);
            l3l[I % PD3] = bld4(wv.w, wv.lane16, (o_A_r3_l + (I + PD3) * 256) * 4);
        }
        w = mfma_h(ah, r2s[c].hi, w);
        w = mfma_h(ah, r2s[c].lo, w);
        w = mfma_h(al, r2s[c].hi, w);
    };
    auto dump_w = [&]<int Tw>(const f32x16& w) {
        if (P.dbg_w != nullptr && valid)
            static_for<16>([&]<int R>() { P.dbg_w[(size_t)e * WN + Tw * 32 + rowmap(R, hi)] = w[R]; });
    };
    DEDF_STAMP(6);
    // prologue: weight tile 0, operands of groups 0 / 1, B operands of group 0
    AOps a_cur = load_A.template operator()<0>();
    XOps x_nxt = load_X.template operator()<1>();
    f32x16 wt = ldrows(wv, o_off_r3, 0);
    BOps b_cur;
    {
        const XOps x0 = load_X.template operator()<0>();
        static_for<4>([&]<int cc>() { sched_fence(); l3_chunk.template operator()<cc>(wt); });
        sched_fence();
        dump_w.template operator()<0>(wt);
        b_cur = valu_group.template operator()<0>(x0, wt);
    }
    DEDF_STAMP(7);
    static_for<NWT>([&]<int Tw>() {
        f32x16 wt_next = wt;
        if constexpr (Tw + 1 < NWT) wt_next = ldrows(wv, o_off_r3, Tw + 1);
        static_for<4>([&]<int g>() {
            constexpr int G = Tw * 4 + g;
            const AOps a_nxt = load_A.template operator()<G + 1>();
            const XOps x_nn = load_X.template operator()<G + 2>();
            sched_fence();
            if constexpr (Tw + 1 < NWT) {       // layer 3 of the next tile: chunks {0,1 | 2 | 3 | -}
                constexpr int first = g == 0 ? 0 : g + 1, cnt = g == 0 ? 2 : (g == 3 ? 0 : 1);
                static_for<cnt>([&]<int k>() { l3_chunk.template operator()<(Tw + 1) * 4 + first + k>(wt_next); });
            }
            const BOps b_nxt = valu_group.template operator()<G + 1>(x_nxt, g == 3 ? wt_next : wt);
            mfma_dtp.template operator()<G>(a_cur, b_cur);
            sched_fence();
            if constexpr (g == 2 && Tw + 1 < NWT) dump_w.template operator()<Tw + 1>(wt_next);
            a_cur = a_nxt; x_nxt = x_nn; b_cur = b_nxt;
        });
        wt = wt_next;
    });

    DEDF_STAMP(8);
    // ---- attention logits (graph_attention.py:233-246): heads of sep_alpha -> SmoothLeakyReLU -> . alpha_dot + log cut-off
    float logit[kHeads];
    {
        constexpr int AT = alpha_row0<L>() / 32;
        static_for<kHeads>([&]<int hd>() {
            constexpr int T = AT + (hd >> 1), r0 = 8 * (hd & 1);
            const f32x4 d0 = bld4(wv.w, wv.hi64, (o_alpha_dot + (hd >> 1) * 32 + r0) * 4);
            const f32x4 d1v = bld4(wv.w, wv.hi64, (o_alpha_dot + (hd >> 1) * 32 + r0 + 4) * 4);
            float s = 0.0f;
            static_for<4>([&]<int R>() { s += slrelu_n(acc[T][r0 + R]) * d0[R]; });
            static_for<4>([&]<int R>() { s += slrelu_n(acc[T][r0 + 4 + R]) * d1v[R]; });
            s += xor32(s);
            logit[hd] = s + logit0;
        });
    }

    // ---- Gate (fast_activation.py:210-224): SiLU on the 64 scalars, sigmoid gates on the l >= 1 channels -------------------
    // u0[T][r]: scalars; u1[m][r] / u2[m][r]: gated l = 1 / l = 2 channels (row = channel)
    f32x16 u0[2];
    static_for<2>([&]<int T>() { static_for<16>([&]<int R>() { u0[T][R] = silu_n(acc[T][R]); }); });
    float u1[3][16], u2[5][8];
    if constexpr (L >= 1) {
        constexpr int G0 = gate_row(1, 0);
        static_for<16>([&]<int R>() {
            const float gt = sigmoid_n(acc[G0 / 32][(G0 % 32) / 2 + R]);
            static_for<3>([&]<int K>() { u1[K][R] = acc[AB1 + K][R] * gt; });
        });
    }
    if constexpr (L >= 2) {
        constexpr int G0 = gate_row(2, 0);
        // 16x16 accumulators back to the row layout (row = channel, lane = edge column + row half): two row exchanges
        static_for<5>([&]<int K>() { static_for<4>([&]<int q>() {
            float x = acc2[K][0][q], y = acc2[K][1][q];
            swap16(x, y);
            swap32(x, y);
            u2[K][q] = x; u2[K][4 + q] = y;
        }); });
        static_for<8>([&]<int R>() {
            const float gt = sigmoid_n(acc[G0 / 32][(G0 % 32) / 2 + R]);
            static_for<5>([&]<int K>() { u2[K][R] *= gt; });
        });
    }
    sched_fence();

    DEDF_STAMP(9);
    // ---- sep_value: depth-wise TP #2 (shared weights folded into A_val) + LinearRS -> value --------------------------------
    constexpr int NV = 2 + (L >= 1 ? 3 : 0);
    f32x16 val[NV];
    f32x4 val2[5][2];
    static_for<5>([&]<int K>() { static_for<2>([&]<int S>() { val2[K][S] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }); });
    static_for<2>([&]<int T>() { val[T] = ldrows(wv, o_b_val0, T); });
    static_for<NV - 2>([&]<int T>() { static_for<16>([&]<int R>() { val[2 + T][R] = 0.0f; }); });
    // groups are walked in weight order (= path creation order, u ascending); region G = loads(G+2), VALU(G+1), MFMA(G)
    struct ValOps { f32x4 a[2]; };
    auto load_val = [&]<int G>() {
        ValOps o;
        if constexpr (G < NGRP) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l3 = pi.l3, gi = dtp_group_index<L>(G), NG = dtp_k<L>(l3) / 8;
            if constexpr (l3 == 2) { const f32x2 t = lda16(wv, o_A_val[2], gi); o.a[0][0] = t[0]; o.a[0][1] = t[1]; }
            else o.a[0] = lda(wv, o_A_val[l3], NG, 0, gi);
            if constexpr (l3 == 0) o.a[1] = lda(wv, o_A_val[0], NG, 1, gi);
        }
        return o;
    };
    auto valu_val = [&]<int G>() {
        BOps o;
        if constexpr (G < NGRP) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d1 = 2 * l1 + 1, d3 = 2 * l3 + 1;
            constexpr int gu = (G * 8 - pi.wstart) / 8;
            using C = CG<l1, l2, l3>;
            float m[C::NM];
            C::make(Y.template get<l2>(), m);
            static_for<4>([&]<int j>() {
                float x[d1], t[d3];
                if constexpr (l1 == 0) x[0] = u0[gu / 4][4 * (gu % 4) + j];
                else if constexpr (l1 == 1) { static_for<3>([&]<int I>() { x[I] = u1[I][4 * gu + j]; }); }
                else { static_for<5>([&]<int I>() { x[I] = u2[I][4 * gu + j]; }); }
                C::apply(x, m, t);
                static_for<d3>([&]<int K>() { o.b[j][K] = t[K]; });
            });
        }
        return o;
    };
    ValOps v0 = load_val.template operator()<0>(), v1 = load_val.template operator()<1>();
    BOps vb_cur = valu_val.template operator()<0>();
    static_for<NGRP>([&]<int G>() {
        constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
        constexpr int l3 = pi.l3, d3 = 2 * l3 + 1;
        const ValOps v2 = load_val.template operator()<G + 2>();
        sched_fence();
        const BOps vb_nxt = valu_val.template operator()<G + 1>();
        if constexpr (l3 == 0) {
            static_for<2>([&]<int To>() { mfma_group(val[To], v0.a[To], vb_cur.b[0][0], vb_cur.b[1][0], vb_cur.b[2][0], vb_cur.b[3][0]); });
        } else if constexpr (l3 == 1) {
            static_for<d3>([&]<int K>() { mfma_group(val[2 + K], v0.a[0], vb_cur.b[0][K], vb_cur.b[1][K], vb_cur.b[2][K], vb_cur.b[3][K]); });
        } else {
            static_for<2>([&]<int pr>() {
                float x[5], y[5];
                static_for<5>([&]<int K>() { x[K] = vb_cur.b[2 * pr][K]; y[K] = vb_cur.b[2 * pr + 1][K]; });
                swap16x5(x, y);
                static_for<5>([&]<int K>() {
                    val2[K][0] = mfma16(v0.a[0][pr], x[K], val2[K][0]);
                    val2[K][1] = mfma16(v0.a[0][pr], y[K], val2[K][1]);
                });
            });
        }
        sched_fence();
        v0 = v1; v1 = v2; vb_cur = vb_nxt;
    });

    DEDF_STAMP(10);
    // ---- store the edge record: value in internal layout [l][m][channel] + one logit per head ----------------------------
    if (valid) {
        float* o = P.out + (size_t)e * REC;
        static_for<2>([&]<int T>() { static_for<4>([&]<int g>() {
            st4(o + T * 32 + 8 * g + 4 * hi, f32x4{val[T][4 * g], val[T][4 * g + 1], val[T][4 * g + 2], val[T][4 * g + 3]});
        }); });
        if constexpr (L >= 1) static_for<3>([&]<int K>() { static_for<4>([&]<int g>() {
            st4(o + blk_off(1) + K * 32 + 8 * g + 4 * hi,
                f32x4{val[2 + K][4 * g], val[2 + K][4 * g + 1], val[2 + K][4 * g + 2], val[2 + K][4 * g + 3]});
        }); });
        if (hi == 0) st4(o + D, f32x4{logit[0], logit[1], logit[2], logit[3]});
    }
    if constexpr (L >= 2) {   // l = 2 block straight from the 16x16 layout: lane (g, e') holds channels 4g..4g+3 of edges e' and 16+e'
        const int g4 = (wv.lane >> 4) * 4, el = wv.lane & 15;
        static_for<2>([&]<int S>() {
            if (el + 16 * S < n_valid) {
                float* o2 = P.out + (size_t)(e0 + el + 16 * S) * REC + blk_off(2) + g4;
                static_for<5>([&]<int K>() { st4(o2 + K * 16, val2[K][S]); });
            }
        });
    }
    DEDF_STAMP(11);
}

}  // namespace dedf
