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
    const float* tb;          // [(nT|1)][n_scales][128] row-packed: W_pre[:,64:] c_t + b_pre
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
    int o_A_pre;              // [n_scales][4 tiles][8 groups][64][4]
    int o_A_r1, o_b_r1, o_g_r1, o_be_r1;
    int o_A_r2, o_b_r2, o_g_r2, o_be_r2;
    int o_A_r3, o_off_r3;
    int o_A_lin[4];           // l3 = 0: lin0 rows + alpha rows; l3 >= 1: mul(l3) rows
    int o_b_r0;               // row-packed bias over the l3 = 0 row space
    int o_A_val[4];           // sep_value.lin with the shared DTP weights folded in
    int o_b_val0;             // row-packed (64)
    int o_alpha_dot;          // row-packed over the two alpha tiles
    float* out;               // [E][edge_rec]
    float* dbg_w;             // optional [E][WN] dump of the radial weights (tests)
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

template <int L>
DEDF_DEV void edge_tile(const EdgeParams& P, const Wave& wv, int scale, int e0, int n_valid) {
    constexpr int D = feat_dim<L>();
    constexpr int REC = edge_rec<L>();
    constexpr int WN = dtp_wn<L>();
    constexpr int NWT = WN / 32;
    constexpr int NR0 = r0_tiles<L>();
    const int hi = wv.hi;
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
                    eb[4 * G + J] = expf(-0.5f * (z * z)) * w[J];
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

    // ---- edge pre-linear + SiLU (multiscale_tensor_field.py:225-234); time part + bias arrive as per-pose rows --------
    f32x16 h[4];
    {
        const Buf tbb = make_buf(P.tb, P.tb_bytes);
        const int tvoff = (pose * P.tb_pose_stride) * 4 + wv.hi64;
        const int oA = P.o_A_pre + scale * (4 * 8 * 256);
        static_for<4>([&]<int To>() { h[To] = ldrows(tbb, tvoff, scale * 128, To); });
        a_stream<32, 4>(wv, [&]<int I>() { return (oA + I * 256) * 4; }, [&]<int I>(f32x4 a) {
            constexpr int To = I / 8, g = I % 8;
            mfma_group(h[To], a, eb[4 * g], eb[4 * g + 1], eb[4 * g + 2], eb[4 * g + 3]);
        });
        static_for<4>([&]<int To>() { static_for<16>([&]<int R>() { h[To][R] = siluf(h[To][R]); }); });
    }
    // ---- RadialProfile layers 1, 2 (equiformer/radial_func.py:11-60) ---------------------------------------------------
    f32x16 r1[4];
    static_for<4>([&]<int To>() { r1[To] = ldrows(wv, P.o_b_r1, To); });
    a_stream<64, 4>(wv, [&]<int I>() { return (P.o_A_r1 + I * 256) * 4; }, [&]<int I>(f32x4 a) {
        constexpr int To = I / 16, T = (I % 16) / 4, g = I % 4;
        mfma_group(r1[To], a, h[T][4 * g], h[T][4 * g + 1], h[T][4 * g + 2], h[T][4 * g + 3]);
    });
    ln_silu<4>(r1, wv, P.o_g_r1, P.o_be_r1);
    f32x16 r2[2];
    static_for<2>([&]<int To>() { r2[To] = ldrows(wv, P.o_b_r2, To); });
    a_stream<32, 4>(wv, [&]<int I>() { return (P.o_A_r2 + I * 256) * 4; }, [&]<int I>(f32x4 a) {
        constexpr int To = I / 16, T = (I % 16) / 4, g = I % 4;
        mfma_group(r2[To], a, r1[T][4 * g], r1[T][4 * g + 1], r1[T][4 * g + 2], r1[T][4 * g + 3]);
    });
    ln_silu<2>(r2, wv, P.o_g_r2, P.o_be_r2);

    // ---- layer 3 (-> per-edge TP weights, one 32-row tile at a time) fused with DTP #1 and the lin / sep_alpha GEMMs ----
    // accumulators: l3 = 0 -> NR0 tiles (lin scalars+gates | alpha), l3 >= 1 -> one tile per m
    constexpr int NACC = NR0 + (L >= 1 ? 3 : 0) + (L >= 2 ? 5 : 0);
    constexpr int AB1 = NR0, AB2 = NR0 + 3;   // first accumulator tile of the l3 = 1 / l3 = 2 outputs
    f32x16 acc[NACC];
    static_for<NR0>([&]<int T>() { acc[T] = ldrows(wv, P.o_b_r0, T); });
    static_for<NACC - NR0>([&]<int T>() { static_for<16>([&]<int R>() { acc[NR0 + T][R] = 0.0f; }); });
    const Buf msgb = make_buf(P.msg, P.msg_bytes);
    // per-l1 lane offsets of the 4 message rows this lane owns inside an 8-row group
    const int mv0 = src * (D * 4) + hi * 16, mv1 = src * (D * 4) + hi * 48, mv2 = src * (D * 4) + hi * 80;

    // operands of one DTP group (8 weight rows): A operands of the lin GEMM(s) it feeds + this lane's 4 source-message rows
    struct GroupOps { f32x4 a[NR0]; f32x4 x[2 * L + 1]; };
    auto load_group = [&]<int G>() {
        GroupOps o;
        if constexpr (G < WN / 8) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l1 = pi.l1, l3 = pi.l3, d1 = 2 * l1 + 1, u0 = G * 8 - pi.wstart;
            constexpr int gi = dtp_group_index<L>(G), NG = dtp_k<L>(l3) / 8;
            if constexpr (l3 == 0) static_for<NR0>([&]<int To>() { o.a[To] = lda(wv, P.o_A_lin[0], NG, To, gi); });
            else o.a[0] = lda(wv, P.o_A_lin[l3], NG, 0, gi);
            const int mv = l1 == 0 ? mv0 : (l1 == 1 ? mv1 : mv2);
            static_for<d1>([&]<int Q>() { o.x[Q] = bld4(msgb, mv, (blk_off(l1) + u0 * d1 + 4 * Q) * 4); });
        }
        return o;
    };
    // layer-3 A operands: one global stream over all tiles, 4 groups (>= 1000 cycles of MFMA work) ahead
    constexpr int NL3 = NWT * 8, PD3 = 4;
    f32x4 l3ring[PD3];
    static_for<PD3>([&]<int I>() { l3ring[I] = bld4(wv.w, wv.lane16, (P.o_A_r3 + I * 256) * 4); });
    GroupOps gcur = load_group.template operator()<0>();
    f32x16 wt = ldrows(wv, P.o_off_r3, 0);
    static_for<NWT>([&]<int Tw>() {
        static_for<8>([&]<int gg>() {
            constexpr int I = Tw * 8 + gg, T = gg / 4, g = gg % 4;
            const f32x4 a = l3ring[I % PD3];
            if constexpr (I + PD3 < NL3) l3ring[I % PD3] = bld4(wv.w, wv.lane16, (P.o_A_r3 + (I + PD3) * 256) * 4);
            sched_fence();
            mfma_group(wt, a, r2[T][4 * g], r2[T][4 * g + 1], r2[T][4 * g + 2], r2[T][4 * g + 3]);
            sched_fence();
        });
        if (P.dbg_w != nullptr && valid)
            static_for<16>([&]<int R>() { P.dbg_w[(size_t)e * WN + Tw * 32 + rowmap(R, hi)] = wt[R]; });
        f32x16 wt_next = wt;
        if constexpr (Tw + 1 < NWT) wt_next = ldrows(wv, P.o_off_r3, Tw + 1);
        static_for<4>([&]<int g>() {
            constexpr int G = Tw * 4 + g;
            constexpr int wrow0 = G * 8;
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(wrow0));
            constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3;
            constexpr int d1 = 2 * l1 + 1, d3 = 2 * l3 + 1;
            const GroupOps gnext = load_group.template operator()<G + 1>();
            sched_fence();
            using C = CG<l1, l2, l3>;
            float m[C::NM];
            C::make(Y.template get<l2>(), m);
            float xr[4 * d1];
            static_for<d1>([&]<int Q>() {
                xr[4 * Q] = gcur.x[Q][0]; xr[4 * Q + 1] = gcur.x[Q][1]; xr[4 * Q + 2] = gcur.x[Q][2]; xr[4 * Q + 3] = gcur.x[Q][3];
            });
            float a[4][d3];
            static_for<4>([&]<int j>() {
                float o[d3];
                C::apply(&xr[j * d1], m, o);
                static_for<d3>([&]<int K>() { a[j][K] = o[K] * wt[4 * g + j]; });
            });
            if constexpr (l3 == 0) {
                static_for<NR0>([&]<int To>() { mfma_group(acc[To], gcur.a[To], a[0][0], a[1][0], a[2][0], a[3][0]); });
            } else {
                static_for<d3>([&]<int K>() {
                    mfma_group(acc[(l3 == 1 ? AB1 : AB2) + K], gcur.a[0], a[0][K], a[1][K], a[2][K], a[3][K]);
                });
            }
            sched_fence();
            gcur = gnext;
        });
        wt = wt_next;
    });

    // ---- attention logits (graph_attention.py:233-246): heads of sep_alpha -> SmoothLeakyReLU -> . alpha_dot + log cut-off
    float logit[kHeads];
    {
        constexpr int AT = alpha_row0<L>() / 32;
        static_for<kHeads>([&]<int hd>() {
            constexpr int T = AT + (hd >> 1), r0 = 8 * (hd & 1);
            const f32x4 d0 = bld4(wv.w, wv.hi64, (P.o_alpha_dot + (hd >> 1) * 32 + r0) * 4);
            const f32x4 d1v = bld4(wv.w, wv.hi64, (P.o_alpha_dot + (hd >> 1) * 32 + r0 + 4) * 4);
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
        static_for<8>([&]<int R>() {
            const float gt = sigmoid_n(acc[G0 / 32][(G0 % 32) / 2 + R]);
            static_for<5>([&]<int K>() { u2[K][R] = acc[AB2 + K][R] * gt; });
        });
    }
    sched_fence();

    // ---- sep_value: depth-wise TP #2 (shared weights folded into A_val) + LinearRS -> value --------------------------------
    constexpr int NV = 2 + (L >= 1 ? 3 : 0) + (L >= 2 ? 5 : 0);
    f32x16 val[NV];
    static_for<2>([&]<int T>() { val[T] = ldrows(wv, P.o_b_val0, T); });
    static_for<NV - 2>([&]<int T>() { static_for<16>([&]<int R>() { val[2 + T][R] = 0.0f; }); });
    // groups are walked in weight order (= path creation order, u ascending); A operands two groups ahead
    struct ValOps { f32x4 a[2]; };
    auto load_val = [&]<int G>() {
        ValOps o;
        if constexpr (G < WN / 8) {
            constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
            constexpr int l3 = pi.l3, gi = dtp_group_index<L>(G), NG = dtp_k<L>(l3) / 8;
            o.a[0] = lda(wv, P.o_A_val[l3], NG, 0, gi);
            if constexpr (l3 == 0) o.a[1] = lda(wv, P.o_A_val[0], NG, 1, gi);
        }
        return o;
    };
    ValOps v0 = load_val.template operator()<0>(), v1 = load_val.template operator()<1>();
    static_for<WN / 8>([&]<int G>() {
        constexpr PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(G * 8));
        constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3;
        constexpr int d1 = 2 * l1 + 1, d3 = 2 * l3 + 1;
        constexpr int gu = (G * 8 - pi.wstart) / 8;
        const ValOps v2 = load_val.template operator()<G + 2>();
        sched_fence();
        using C = CG<l1, l2, l3>;
        float m[C::NM];
        C::make(Y.template get<l2>(), m);
        float a[4][d3];
        static_for<4>([&]<int j>() {
            float x[d1], o[d3];
            if constexpr (l1 == 0) x[0] = u0[gu / 4][4 * (gu % 4) + j];
            else if constexpr (l1 == 1) { static_for<3>([&]<int I>() { x[I] = u1[I][4 * gu + j]; }); }
            else { static_for<5>([&]<int I>() { x[I] = u2[I][4 * gu + j]; }); }
            C::apply(x, m, o);
            static_for<d3>([&]<int K>() { a[j][K] = o[K]; });
        });
        if constexpr (l3 == 0) {
            static_for<2>([&]<int To>() { mfma_group(val[To], v0.a[To], a[0][0], a[1][0], a[2][0], a[3][0]); });
        } else {
            static_for<d3>([&]<int K>() { mfma_group(val[(l3 == 1 ? 2 : 5) + K], v0.a[0], a[0][K], a[1][K], a[2][K], a[3][K]); });
        }
        sched_fence();
        v0 = v1; v1 = v2;
    });

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
        if constexpr (L >= 2) static_for<5>([&]<int K>() { static_for<2>([&]<int g>() {
            st4(o + blk_off(2) + K * 16 + 8 * g + 4 * hi,
                f32x4{val[5 + K][4 * g], val[5 + K][4 * g + 1], val[5 + K][4 * g + 2], val[5 + K][4 * g + 3]});
        }); });
        if (hi == 0) st4(o + D, f32x4{logit[0], logit[1], logit[2], logit[3]});
    }
}

}  // namespace dedf
