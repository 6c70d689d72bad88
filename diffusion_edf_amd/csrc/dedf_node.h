// Per-query-node epilogue (one wave = 32 (pose, query) nodes):
//   attention output -> proj (graph_attention.py:268) -> post-norm + FFN + residual (gnn_block.py:210-216) ->
//   lin/ang score tensor products with the Wigner-rotated query feature (score_head.py:192-199) -> rotate back to the
//   body frame, orbital term, query weights (score_head.py:201-209).
// Same transposed-GEMM register layout as the edge kernel (dedf_layout.h): lane = (node column, row half).
//
// The score TP `uvu` with mul(in2) > 1 is evaluated factorised:  T_p[u][j] = sum_v W_p[u,v] field[v,j]  (MFMA),
// then the Clebsch-Gordan contraction with the query feature row u is lane-local and its result row is directly the
// K-step operand of the final LinearRS — 25 088 MAC per TP instead of the 171 632 of the materialised (u,v,i,j) form.
#pragma once
#include "dedf_dev.h"
#include "dedf_net.h"

namespace dedf {

constexpr int kPoseRec = 64;      // floats per pose (lmax <= 2; dedf_net.h::pose_rec<L>()): [0:4] raw q, [4:13] D^1 row-major, [16:41] D^2 row-major, lmax 3: [48:97] D^3

struct NodeParams {
    const float* z;  uint32_t z_bytes;        // [N_d][D] softmax-aggregated values, internal layout
    const float* qf; uint32_t qf_bytes;       // [nQ][D] query features, reference layout
    const float* pose; uint32_t pose_bytes;   // [nT][kPoseRec]
    const float* qx;                          // [nQ][3] query positions (gripper frame)
    const float* qw;                          // [nQ] query weights
    int nQ, n_nodes;
    float lin_mult;
    const float* W; uint32_t W_bytes;
    int o_A_proj[4], o_b_proj0;
    int o_ln_w[4], o_ln_b0;
    int o_A_f1[4], o_b_f1;
    int o_A_f2[4], o_b_f2;
    int o_A_s[2][16];                          // [tp][path]  W_p (rows u, K = v)
    int o_A_sl[2][2];                          // [tp][l3]    final LinearRS (l3 = 0: the 32 gate rows; l3 = 1: 32 rows)
    int o_b_sl[2];                             // [tp]        gate bias rows
    int o_A_proj_l[4], o_A_f1_l[4], o_A_f2_l[4], o_A_s_l[2][16], o_A_sl_l[2][2];     // residual (lo) images of the split-fp16 operands
    NodeScales sc;                             // accumulator -> true value, per matrix
    const float* f_dst; uint32_t f_dst_bytes;  // UNet layer only: [N_d][D] destination input features (first skip connection, block.py:165)
    float* feat_out;                           // UNet layer only: [N_d][D] output features, reference layout
    float ln_inv_n[4], ln_pad0;                // UNet layer only: 1 / (true multiplicity) per degree and the number of padded 0e channels (masked norm_2)
    float* node_out;                           // [N_d][8]: w*lin_vel (3), w*(ang_orbital + ang_spin) (3), 0, 0
    // Small batches (round 5): a node tile's latency is the whole kernel, and its two score tensor products are independent.  With `split` the grid
    // holds TWO waves per tile; both run the shared front (proj, LayerNorm, FFN), wave parity 0 the lin_vel product -> node_out = w*lin_vel,
    // w*ang_orbital; parity 1 the ang_vel product -> node_spin = w*ang_spin.  The reductions add the two.
    int split;
    float* node_spin;                          // [N_d][4]
    const float* skip1; int skip1_stride;      // query_time_encoding: time rows of dedf_misc.h::k_time_query (their second, row-packed half): skip_1(destination
                                               // feature) joins the 0e block of the attention output (gnn_block.py:111, 205-206); stride in floats between poses (0: shared)
    float* dbg_emb;                            // optional [N_d][D] dumps (internal layout) of the proj output and of the field (tests)
    float* dbg_field;
};

template <int L> struct Feat {                 // one node's features in row layout: row = channel
    f32x16 s[2];                               // 64 scalars
    float v1[3][16];                           // 32x1e, [m][reg]
    float v2[5][8];                            // 16x2e, [m][reg]
    float v3[7][8];                            // 8x3e zero-padded to 16 channels (dedf_net.h::pad_pos), [m][reg]
};
// the same features as split-fp16 B operands: chunks of 16 channels (8 registers of a row-layout tile), scaled by 2^kNodeBShift
template <int L> struct FeatH { HL s[4], v1[3][2], v2[5][1], v3[7][1]; };
template <int L, bool HP = false> DEDF_DEV FeatH<L> split_feat(const Feat<L>& f, const float sc) {
    FeatH<L> o;
    static_for<4>([&]<int c>() { float t[8]; static_for<8>([&]<int J>() { t[J] = f.s[c / 2][8 * (c % 2) + J]; }); o.s[c] = split8sx<HP>(t, sc); });
    if constexpr (L >= 1) static_for<3>([&]<int m>() { static_for<2>([&]<int c>() {
        float t[8]; static_for<8>([&]<int J>() { t[J] = f.v1[m][8 * c + J]; }); o.v1[m][c] = split8sx<HP>(t, sc); }); });
    if constexpr (L >= 2) static_for<5>([&]<int m>() {
        float t[8]; static_for<8>([&]<int J>() { t[J] = f.v2[m][J]; }); o.v2[m][0] = split8sx<HP>(t, sc); });
    if constexpr (L >= 3) static_for<7>([&]<int m>() {
        float t[8]; static_for<8>([&]<int J>() { t[J] = f.v3[m][J]; }); o.v3[m][0] = split8sx<HP>(t, sc); });
    return o;
}

// per-row vectors (biases, LayerNorm affine) of the node image, copied once per wave into LDS (see dedf_edge.h::RowsLds)
template <int L> struct NodeRowsLds {
    static constexpr int b_proj0 = 0, ln_w0 = 64, ln_w1 = 128, ln_w2 = 160, ln_b0 = 192, ln_w3 = 256, b_f1 = 288;
    static constexpr int b_f2 = b_f1 + cdiv(f1_rows0<L>(), 32) * 32, b_sl0 = b_f2 + 64, b_sl1 = b_sl0 + 32, total = b_sl1 + 32;
};
template <int L> DEDF_DEV float* node_rows_lds() {
    __shared__ __attribute__((aligned(16))) float rows[NodeRowsLds<L>::total];
    return rows;
}
template <int L, bool EBM>
DEDF_DEV void node_rows_to_lds(const NodeParams& P, const Wave& wv) {
    using RL = NodeRowsLds<L>;
    constexpr NodeLayout<L> O = kNodeLayout<L>;
    float* rows = node_rows_lds<L>();
    // (all requests, then all stores: dedf_dev.h::rows_request)
    const float* const W = P.W;
    const int lane = wv.lane;
    constexpr int NF1 = cdiv(f1_rows0<L>(), 32) * 32;
    const RowRegs<64> bp = rows_request<64>(W + O.b_proj0, lane), w0 = rows_request<64>(W + O.ln_w[0], lane), lb = rows_request<64>(W + O.ln_b0, lane),
                      bf2 = rows_request<64>(W + O.b_f2, lane);
    RowRegs<32> w1, w2, w3, s0, s1;
    if constexpr (L >= 1) w1 = rows_request<32>(W + O.ln_w[1], lane);
    if constexpr (L >= 2) w2 = rows_request<32>(W + O.ln_w[2], lane);
    if constexpr (L >= 3) w3 = rows_request<32>(W + O.ln_w[3], lane);
    const RowRegs<NF1> bf1 = rows_request<NF1>(W + O.b_f1, lane);
    if constexpr (!EBM) { s0 = rows_request<32>(W + O.b_sl[0], lane); s1 = rows_request<32>(W + O.b_sl[1], lane); }
    sched_fence();
    rows_store<64>(rows + RL::b_proj0, bp, lane); rows_store<64>(rows + RL::ln_w0, w0, lane); rows_store<64>(rows + RL::ln_b0, lb, lane);
    if constexpr (L >= 1) rows_store<32>(rows + RL::ln_w1, w1, lane);
    if constexpr (L >= 2) rows_store<32>(rows + RL::ln_w2, w2, lane);
    if constexpr (L >= 3) rows_store<32>(rows + RL::ln_w3, w3, lane);
    rows_store<NF1>(rows + RL::b_f1, bf1, lane); rows_store<64>(rows + RL::b_f2, bf2, lane);
    if constexpr (!EBM) { rows_store<32>(rows + RL::b_sl0, s0, lane); rows_store<32>(rows + RL::b_sl1, s1, lane); }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
DEDF_DEV f32x16 node_ldrows(const float* rows, int hi, int off, int tile) {
    f32x16 v;
    const f32x4* p = reinterpret_cast<const f32x4*>(rows + off + tile * 32 + hi * 16);
    static_for<4>([&]<int G>() { const f32x4 t = p[G]; v[4 * G] = t[0]; v[4 * G + 1] = t[1]; v[4 * G + 2] = t[2]; v[4 * G + 3] = t[3]; });
    return v;
}

// UN: the node half of a UNet layer (block.py:164-172): out1 = f_dst + proj(attention); out = out1 + FFN(norm_2(out1)) -> feat_out.
// (the `emb` of the score head plays out1: its residual `field = FFN(LN(emb)) + emb` is exactly that second skip connection)
// Feature rows in the reference layout <-> row layout: lane (hi) owns channels 8 gu + 4 hi + j (j < 4) of every 8-channel group gu,
// i.e. 4 (2l+1) contiguous floats of a node's block l.
template <int L> DEDF_DEV void feat_add_ref(Feat<L>& f, const Buf& b, int node_off_bytes, int hi) {
    static_for<L + 1>([&]<int l>() {
        constexpr int d = 2 * l + 1;
        const int v = node_off_bytes + hi * (16 * d);
        static_for<mul_of(l) / 8>([&]<int gu>() {
            float xr[4 * d];
            static_for<d>([&]<int Q>() {
                const f32x4 t = bld4(b, v, (blk_off(l) + gu * 8 * d + 4 * Q) * 4);
                xr[4 * Q] = t[0]; xr[4 * Q + 1] = t[1]; xr[4 * Q + 2] = t[2]; xr[4 * Q + 3] = t[3];
            });
            static_for<4>([&]<int j>() { static_for<d>([&]<int I>() {
                if constexpr (l == 0) f.s[gu / 4][4 * (gu % 4) + j] += xr[j];
                else if constexpr (l == 1) f.v1[I][4 * gu + j] += xr[j * d + I];
                else if constexpr (l == 2) f.v2[I][4 * gu + j] += xr[j * d + I];
                else f.v3[I][4 * gu + j] += xr[j * d + I];
            }); });
        });
    });
}
template <int L> DEDF_DEV void feat_store_ref(const Feat<L>& f, float* node_ptr, int hi) {
    static_for<L + 1>([&]<int l>() {
        constexpr int d = 2 * l + 1;
        float* const o = node_ptr + blk_off(l) + hi * (4 * d);
        static_for<mul_of(l) / 8>([&]<int gu>() {
            float xr[4 * d];
            static_for<4>([&]<int j>() { static_for<d>([&]<int I>() {
                if constexpr (l == 0) xr[j] = f.s[gu / 4][4 * (gu % 4) + j];
                else if constexpr (l == 1) xr[j * d + I] = f.v1[I][4 * gu + j];
                else if constexpr (l == 2) xr[j * d + I] = f.v2[I][4 * gu + j];
                else xr[j * d + I] = f.v3[I][4 * gu + j];
            }); });
            static_for<d>([&]<int Q>() { st4(o + gu * 8 * d + 4 * Q, f32x4{xr[4 * Q], xr[4 * Q + 1], xr[4 * Q + 2], xr[4 * Q + 3]}); });
        });
    });
}

// the (path, row tile) products of one score tensor product in the order node_tile walks them: grouped by input degree (the rotated query rows of a
// degree serve all its paths), then path, then row tile -- so that every product can request the first operands of the NEXT one
struct StpStep { int p, To; };
template <int L> DEDF_HD constexpr int stp_num_steps() {
    int n = 0;
    for (int p = 0; p < stp_num_paths<L>(); ++p) n += cdiv(stp_path<L>(p).mul1, 32);
    return n;
}
template <int L> DEDF_HD constexpr StpStep stp_step(int i) {
    int n = 0;
    for (int l1 = 0; l1 <= L; ++l1)
        for (int p = 0; p < stp_num_paths<L>(); ++p) {
            if (stp_path<L>(p).l1 != l1) continue;
            for (int To = 0; To < cdiv(stp_path<L>(p).mul1, 32); ++To) { if (n == i) return StpStep{p, To}; ++n; }
        }
    return StpStep{-1, 0};
}
template <int L> DEDF_HD constexpr int stp_step_index(int p, int To) {
    for (int i = 0; i < stp_num_steps<L>(); ++i) if (stp_step<L>(i).p == p && stp_step<L>(i).To == To) return i;
    return -1;
}

template <int L, bool EBM, bool HP = false, bool UN = false>
DEDF_DEV void node_tile(const NodeParams& P, const Wave& wv, int n0, int tp_only = -1) {
    constexpr int D = feat_dim<L>();
    const int hi = wv.hi;
    constexpr NodeLayout<L> O = kNodeLayout<L>;      // weight-image offsets: compile-time constants (dedf_net.h)
    using NR = NodeRowsLds<L>;
    const float* const rows = node_rows_lds<L>();
    const bool valid = n0 + wv.col < P.n_nodes;
    const int n = valid ? n0 + wv.col : n0;
    const int pose = n / P.nQ, q = n - pose * P.nQ;

#ifndef DEDF_NODE_CARRY
#define DEDF_NODE_CARRY 2    // 1: every (path, row tile) product of the score tensor products requests the first operands of the next one before it starts;
                             // 2: the shared-operand products of proj / FFN too;  3: and their rotated-accumulator products (A/B: 0)
#endif
    constexpr bool CARRY2 = DEDF_NODE_CARRY >= 2 && L <= 2;      // (lmax 3: 300 -> 360 B of scratch with it)
    constexpr bool CARRY3 = DEDF_NODE_CARRY >= 3 && L <= 2;
    using SR = SharedRing<2>;
    auto pf = [&]<int NCH>(int off_h, int off_l, int nCH, int To) -> SR {      // first operands of a later product, requested now
        if constexpr (CARRY2) return shared_prefetch<NCH, 2, HP>(wv, off_h, off_l, nCH, To); else return SR{};
    };
    auto gemm_sh = [&]<int NM, int NCH>(int off_h, int off_l, int nCH, int To, f32x16 (&acc)[NM], auto&& bh, const SR& ring) {
        if constexpr (CARRY2) dense_shared_hp<NM, NCH, 2, HP>(wv, off_h, off_l, nCH, To, acc, bh, ring);
        else dense_shared_hp<NM, NCH, 2, HP>(wv, off_h, off_l, nCH, To, acc, bh);
    };
    // ---- load aggregated attention values (internal layout [l][m][channel]) ---------------------------------------------
    Feat<L> z;
    {
        const Buf zb = make_buf(P.z, P.z_bytes);
        const int zv = n * (D * 4) + hi * 16;
        static_for<2>([&]<int T>() { static_for<4>([&]<int g>() {
            const f32x4 t = bld4(zb, zv, (T * 32 + 8 * g) * 4);
            z.s[T][4 * g] = t[0]; z.s[T][4 * g + 1] = t[1]; z.s[T][4 * g + 2] = t[2]; z.s[T][4 * g + 3] = t[3];
        }); });
        if constexpr (L >= 1) static_for<3>([&]<int m>() { static_for<4>([&]<int g>() {
            const f32x4 t = bld4(zb, zv, (blk_off(1) + m * 32 + 8 * g) * 4);
            z.v1[m][4 * g] = t[0]; z.v1[m][4 * g + 1] = t[1]; z.v1[m][4 * g + 2] = t[2]; z.v1[m][4 * g + 3] = t[3];
        }); });
        if constexpr (L >= 2) static_for<5>([&]<int m>() { static_for<2>([&]<int g>() {
            const f32x4 t = bld4(zb, zv, (blk_off(2) + m * 16 + 8 * g) * 4);
            z.v2[m][4 * g] = t[0]; z.v2[m][4 * g + 1] = t[1]; z.v2[m][4 * g + 2] = t[2]; z.v2[m][4 * g + 3] = t[3];
        }); });
        if constexpr (L >= 3) static_for<7>([&]<int m>() { static_for<2>([&]<int g>() {
            const f32x4 t = bld4(zb, zv, (blk_off(3) + m * 16 + 8 * g) * 4);
            z.v3[m][4 * g] = t[0]; z.v3[m][4 * g + 1] = t[1]; z.v3[m][4 * g + 2] = t[2]; z.v3[m][4 * g + 3] = t[3];
        }); });
    }

    // ---- proj: per-l dense matrix (+ bias on 0e); all GEMMs of this kernel are 3-term split-fp16 MFMA products -----------------
    Feat<L> emb;
    {
        SR rp1{}, rp2{}, rp3{};
        DenseRing<2, 2> rp0{};
        if constexpr (CARRY3) rp0 = dense_prefetch<2, 4, 2, HP>(wv, O.A_proj[0], O.A_proj_l[0]);
        if constexpr (L >= 1) rp1 = pf.template operator()<2>(O.A_proj[1], O.A_proj_l[1], 2, 0);
        const FeatH<L> zh = split_feat<L, HP>(z, opaque_s(P.sc.bz));
        f32x16 a0[2];
        static_for<2>([&]<int To>() { a0[To] = node_ldrows(rows, hi, NR::b_proj0, To); });
        if constexpr (CARRY3) dense_rot_hp<2, 4, 2, HP>(wv, O.A_proj[0], O.A_proj_l[0], a0, [&]<int c>() { return zh.s[c]; }, rp0);
        else dense_rot_hp<2, 4, 2, HP>(wv, O.A_proj[0], O.A_proj_l[0], a0, [&]<int c>() { return zh.s[c]; });
        const float c0 = opaque_s(P.sc.proj[0]);
        static_for<2>([&]<int To>() { static_for<16>([&]<int R>() { emb.s[To][R] = a0[To][R] * c0; }); });
        if constexpr (L >= 1) {
            f32x16 a[3] = {{0}, {0}, {0}};
            if constexpr (L >= 2) rp2 = pf.template operator()<1>(O.A_proj[2], O.A_proj_l[2], 1, 0);
            gemm_sh.template operator()<3, 2>(O.A_proj[1], O.A_proj_l[1], 2, 0, a, [&]<int m, int c>() { return zh.v1[m][c]; }, rp1);
            const float c1 = opaque_s(P.sc.proj[1]);
            static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { emb.v1[m][R] = a[m][R] * c1; }); });
        }
        if constexpr (L >= 2) {
            f32x16 a[5] = {{0}, {0}, {0}, {0}, {0}};
            if constexpr (L >= 3) rp3 = pf.template operator()<1>(O.A_proj[3], O.A_proj_l[3], 1, 0);
            gemm_sh.template operator()<5, 1>(O.A_proj[2], O.A_proj_l[2], 1, 0, a, [&]<int m, int c>() { return zh.v2[m][c]; }, rp2);
            const float c2 = opaque_s(P.sc.proj[2]);
            static_for<5>([&]<int m>() { static_for<8>([&]<int R>() { emb.v2[m][R] = a[m][R] * c2; }); });
        }
        if constexpr (L >= 3) {
            f32x16 a[7] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}};
            gemm_sh.template operator()<7, 1>(O.A_proj[3], O.A_proj_l[3], 1, 0, a, [&]<int m, int c>() { return zh.v3[m][c]; }, rp3);
            const float c3 = opaque_s(P.sc.proj[3]);
            static_for<7>([&]<int m>() { static_for<8>([&]<int R>() { emb.v3[m][R] = a[m][R] * c3; }); });
        }
    }

    if constexpr (UN) feat_add_ref<L>(emb, make_buf(P.f_dst, P.f_dst_bytes), n * (D * 4), hi);      // node_output = node_input_dst + ga(...)
    if constexpr (!UN) if (P.skip1 != nullptr) {      // query_time_encoding: + skip_1(query_time_mlp(time of the pose))
        const f32x4* const r = reinterpret_cast<const f32x4*>(P.skip1 + (size_t)pose * P.skip1_stride + hi * 16);
        static_for<2>([&]<int T>() { static_for<4>([&]<int G>() {
            const f32x4 t = r[T * 8 + G];
            emb.s[T][4 * G] += t[0]; emb.s[T][4 * G + 1] += t[1]; emb.s[T][4 * G + 2] += t[2]; emb.s[T][4 * G + 3] += t[3];
        }); });
    }

    // ---- EquivariantLayerNormV2 (equiformer/layer_norm.py:91-156) ----------------------------------------------------------
    Feat<L> nrm;
    {
        float s = 0.0f;
        static_for<2>([&]<int T>() { static_for<16>([&]<int R>() { s += emb.s[T][R]; }); });
        s += xor32(s);
        // (UN: masked statistics over the true channels only, see dedf_edge.h::ln_silu)
        const float mean = UN ? s * P.ln_inv_n[0] : s * (1.0f / 64);
        float v = 0.0f;
        static_for<2>([&]<int T>() { static_for<16>([&]<int R>() { const float d = emb.s[T][R] - mean; v += d * d; }); });
        v += xor32(v);
        if constexpr (UN) v = fmaxf(v - P.ln_pad0 * (mean * mean), 0.0f);
        const float rs = 1.0f / sqrtf(v * (UN ? P.ln_inv_n[0] : 1.0f / 64) + 1e-5f);
        static_for<2>([&]<int T>() {
            const f32x16 w = node_ldrows(rows, hi, NR::ln_w0, T), b = node_ldrows(rows, hi, NR::ln_b0, T);
            static_for<16>([&]<int R>() { nrm.s[T][R] = (emb.s[T][R] - mean) * (rs * w[R]) + b[R]; });
        });
    }
    if constexpr (L >= 1) {
        float v = 0.0f;
        static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { v += emb.v1[m][R] * emb.v1[m][R]; }); });
        v += xor32(v);
        const float rs = 1.0f / sqrtf(v * (UN ? P.ln_inv_n[1] * (1.0f / 3) : 1.0f / (3 * 32)) + 1e-5f);
        const f32x16 w = node_ldrows(rows, hi, NR::ln_w1, 0);
        static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { nrm.v1[m][R] = emb.v1[m][R] * (rs * w[R]); }); });
    }
    if constexpr (L >= 2) {
        float v = 0.0f;
        static_for<5>([&]<int m>() { static_for<8>([&]<int R>() { v += emb.v2[m][R] * emb.v2[m][R]; }); });
        v += xor32(v);
        const float rs = 1.0f / sqrtf(v * (UN ? P.ln_inv_n[2] * (1.0f / 5) : 1.0f / (5 * 16)) + 1e-5f);
        const f32x16 w = node_ldrows(rows, hi, NR::ln_w2, 0);
        static_for<5>([&]<int m>() { static_for<8>([&]<int R>() { nrm.v2[m][R] = emb.v2[m][R] * (rs * w[R]); }); });
    }
    if constexpr (L >= 3) {      // the padded channels are exactly 0: the sum is over the true ones, the mean takes the true count
        float v = 0.0f;
        static_for<7>([&]<int m>() { static_for<8>([&]<int R>() { v += emb.v3[m][R] * emb.v3[m][R]; }); });
        v += xor32(v);
        const float rs = 1.0f / sqrtf(v * (UN ? P.ln_inv_n[3] * (1.0f / 7) : 1.0f / (7 * true_mul(3))) + 1e-5f);
        const f32x16 w = node_ldrows(rows, hi, NR::ln_w3, 0);
        static_for<7>([&]<int m>() { static_for<8>([&]<int R>() { nrm.v3[m][R] = emb.v3[m][R] * (rs * w[R]); }); });
    }

    // ---- FFN: FCTP+SwishGate (D -> 336x0e+96x1e+48x2e) -> Gate -> FCTP (-> D), + residual (gnn_block.py:51-57, 210-216) ----
    Feat<L> fld;
    constexpr int NF1 = f1_rows0<L>() / 32 + (f1_rows0<L>() % 32 ? 1 : 0);     // 11 (L=2) / 9 (L=1) tiles of fctp_1's 0e rows
    // split-fp16 B operands that are reused by several GEMMs are parked in LDS (this wave's 30 KB, slot = one h8 per lane):
    // first the normalised features (FFN), later the field (score tensor products); the registers go to the accumulators
    constexpr int FS1 = 8, FS2 = FS1 + 12, FS3 = FS2 + 10;
    __shared__ f32x4 fpark[(L >= 3 ? FS3 + 14 : FS2 + 10) * 64];
    f32x4* const fp = fpark + wv.lane;
    auto park = [&](const Feat<L>& f, const float bscale) {
        const FeatH<L> fh = split_feat<L, HP>(f, bscale);
        static_for<4>([&]<int c>() { fp[(2 * c) * 64] = __builtin_bit_cast(f32x4, fh.s[c].hi); fp[(2 * c + 1) * 64] = __builtin_bit_cast(f32x4, fh.s[c].lo); });
        if constexpr (L >= 1) static_for<3>([&]<int m>() { static_for<2>([&]<int c>() {
            fp[(FS1 + 4 * m + 2 * c) * 64] = __builtin_bit_cast(f32x4, fh.v1[m][c].hi);
            fp[(FS1 + 4 * m + 2 * c + 1) * 64] = __builtin_bit_cast(f32x4, fh.v1[m][c].lo);
        }); });
        if constexpr (L >= 2) static_for<5>([&]<int m>() {
            fp[(FS2 + 2 * m) * 64] = __builtin_bit_cast(f32x4, fh.v2[m][0].hi);
            fp[(FS2 + 2 * m + 1) * 64] = __builtin_bit_cast(f32x4, fh.v2[m][0].lo);
        });
        if constexpr (L >= 3) static_for<7>([&]<int m>() {
            fp[(FS3 + 2 * m) * 64] = __builtin_bit_cast(f32x4, fh.v3[m][0].hi);
            fp[(FS3 + 2 * m + 1) * 64] = __builtin_bit_cast(f32x4, fh.v3[m][0].lo);
        });
    };
    auto parked = [&]<int l, int m, int c>() {       // chunk c of component m of block l
        constexpr int slot = l == 0 ? 2 * c : (l == 1 ? FS1 + 4 * m + 2 * c : (l == 2 ? FS2 + 2 * m : FS3 + 2 * m));
        HL b;
        b.hi = __builtin_bit_cast(h8, fp[slot * 64]);
        b.lo = __builtin_bit_cast(h8, fp[(slot + 1) * 64]);
        return b;
    };
    DenseRing<6, 1> rf10{};
    if constexpr (CARRY3) rf10 = dense_prefetch<6, 4, 1, HP>(wv, O.A_f1[0], O.A_f1_l[0]);      // under the split + parking of the normalised features
    park(nrm, opaque_s(P.sc.bn));
    const float bh0 = opaque_s(P.sc.bh[0]), bh1 = opaque_s(P.sc.bh[L >= 1 ? 1 : 0]), bh2 = opaque_s(P.sc.bh[L >= 2 ? 2 : 0]), bh3 = opaque_s(P.sc.bh[L >= 3 ? 3 : 0]);
    (void)bh1; (void)bh2; (void)bh3;
    constexpr int NGT0 = NF1 - 6;
    DenseRing<NGT0, 1> rg{};      // first operands of the gate rows' product: requested when the hidden scalars' registers are free (behind fctp_2's l = 0 product)
    {   // l = 0: 192 hidden scalars (6 tiles) -> SiLU -> fctp_2
        f32x16 hs[6];
        static_for<6>([&]<int To>() { hs[To] = node_ldrows(rows, hi, NR::b_f1, To); });
        DenseRing<2, 2> rf20{};
        if constexpr (CARRY3) rf20 = dense_prefetch<2, 12, 2, HP>(wv, O.A_f2[0], O.A_f2_l[0]);
        if constexpr (CARRY3) dense_rot_hp<6, 4, 1, HP>(wv, O.A_f1[0], O.A_f1_l[0], hs, [&]<int c>() { return parked.template operator()<0, 0, c>(); }, rf10);
        else dense_rot_hp<6, 4, 1, HP>(wv, O.A_f1[0], O.A_f1_l[0], hs, [&]<int c>() { return parked.template operator()<0, 0, c>(); });
        const float c1 = opaque_s(P.sc.f1[0]);
        static_for<6>([&]<int To>() { static_for<16>([&]<int R>() { hs[To][R] = silu_n(hs[To][R] * c1); }); });
        f32x16 o0[2];
        static_for<2>([&]<int T>() { o0[T] = node_ldrows(rows, hi, NR::b_f2, T); });
        auto bh_f2 = [&]<int c>() {
            float t[8];
            static_for<8>([&]<int J>() { t[J] = hs[c / 2][8 * (c % 2) + J]; });
            return split8sx<HP>(t, bh0);
        };
        if constexpr (CARRY3) dense_rot_hp<2, 12, 2, HP>(wv, O.A_f2[0], O.A_f2_l[0], o0, bh_f2, rf20);
        else dense_rot_hp<2, 12, 2, HP>(wv, O.A_f2[0], O.A_f2_l[0], o0, bh_f2);
        if constexpr (CARRY3) rg = dense_prefetch<NGT0, 4, 1, HP>(wv, O.A_f1[0] + 6 * 4 * 256, O.A_f1_l[0] + 6 * 4 * 256);
        const float c2 = opaque_s(P.sc.f2[0]);
        static_for<2>([&]<int T>() { static_for<16>([&]<int R>() { fld.s[T][R] = o0[T][R] * c2 + emb.s[T][R]; }); });
    }
    // gate rows of fctp_1 (tiles 6 .. NF1-1 of the 0e row space): 96 gates for the 1e hidden, 48 for the 2e hidden
    constexpr int NGT = NF1 - 6;
    f32x16 gt[NGT];
    static_for<NGT>([&]<int t>() { gt[t] = node_ldrows(rows, hi, NR::b_f1, 6 + t); });
    SR rf1{};       // carried from product to product through the rest of the FFN
    if constexpr (L >= 1) rf1 = pf.template operator()<2>(O.A_f1[1], O.A_f1_l[1], 2, 0);
    {
        // the gate tiles start at tile 6 of the same matrix (4 chunks per tile) -> shift the image offsets
        if constexpr (CARRY3) dense_rot_hp<NGT, 4, 1, HP>(wv, O.A_f1[0] + 6 * 4 * 256, O.A_f1_l[0] + 6 * 4 * 256, gt, [&]<int c>() { return parked.template operator()<0, 0, c>(); }, rg);
        else
        dense_rot_hp<NGT, 4, 1, HP>(wv, O.A_f1[0] + 6 * 4 * 256, O.A_f1_l[0] + 6 * 4 * 256, gt, [&]<int c>() { return parked.template operator()<0, 0, c>(); });
        const float c1 = opaque_s(P.sc.f1[0]);
        static_for<NGT>([&]<int t>() { static_for<16>([&]<int R>() { gt[t][R] = sigmoid_n(gt[t][R] * c1); }); });
    }
    if constexpr (L >= 1) {   // l = 1: hidden 96x1e (3 tiles) per component, gated, then fctp_2 (K = 96) shared over the 3 components
        f32x16 hh[3][3];      // [tile][m]
        const float c1 = opaque_s(P.sc.f1[1]);
        static_for<3>([&]<int t>() {
            static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { hh[t][m][R] = 0.0f; }); });
            const SR cur = rf1;
            if constexpr (t < 2) rf1 = pf.template operator()<2>(O.A_f1[1], O.A_f1_l[1], 2, t + 1);
            else rf1 = pf.template operator()<6>(O.A_f2[1], O.A_f2_l[1], 6, 0);
            gemm_sh.template operator()<3, 2>(O.A_f1[1], O.A_f1_l[1], 2, t, hh[t], [&]<int m, int c>() { return parked.template operator()<1, m, c>(); }, cur);
            static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { hh[t][m][R] *= gt[t][R] * c1; }); });
        });
        f32x16 o[3] = {{0}, {0}, {0}};
        const SR cur1 = rf1;
        if constexpr (L >= 2) rf1 = pf.template operator()<1>(O.A_f1[2], O.A_f1_l[2], 1, 0);
        gemm_sh.template operator()<3, 6>(O.A_f2[1], O.A_f2_l[1], 6, 0, o, [&]<int m, int c>() {
            float t[8];
            static_for<8>([&]<int J>() { t[J] = hh[c / 2][m][8 * (c % 2) + J]; });
            return split8sx<HP>(t, bh1);
        }, cur1);
        const float c2 = opaque_s(P.sc.f2[1]);
        static_for<3>([&]<int m>() { static_for<16>([&]<int R>() { fld.v1[m][R] = o[m][R] * c2 + emb.v1[m][R]; }); });
    }
    if constexpr (L >= 2) {   // l = 2: hidden 48x2e (tile 0 full, tile 1 rows 0..15); gates = 0e rows 288..335 (gate tile 3, tile 4 low half)
        f32x16 hh[2][5];
        const float c1 = opaque_s(P.sc.f1[2]);
        static_for<2>([&]<int t>() {
            static_for<5>([&]<int m>() { static_for<16>([&]<int R>() { hh[t][m][R] = 0.0f; }); });
            const SR cur = rf1;
            if constexpr (t < 1) rf1 = pf.template operator()<1>(O.A_f1[2], O.A_f1_l[2], 1, t + 1);
            else rf1 = pf.template operator()<3>(O.A_f2[2], O.A_f2_l[2], 3, 0);
            gemm_sh.template operator()<5, 1>(O.A_f1[2], O.A_f1_l[2], 1, t, hh[t], [&]<int m, int c>() { return parked.template operator()<2, m, c>(); }, cur);
            static_for<5>([&]<int m>() { static_for<16>([&]<int R>() { hh[t][m][R] *= gt[3 + t][R] * c1; }); });
        });
        f32x16 o[5] = {{0}, {0}, {0}, {0}, {0}};
        // K = 48: chunks 0, 1 read hidden tile 0, chunk 2 the valid half of tile 1
        const SR cur2 = rf1;
        if constexpr (L >= 3) rf1 = pf.template operator()<1>(O.A_f1[3], O.A_f1_l[3], 1, 0);
        gemm_sh.template operator()<5, 3>(O.A_f2[2], O.A_f2_l[2], 3, 0, o, [&]<int m, int c>() {
            float t[8];
            static_for<8>([&]<int J>() { t[J] = hh[c / 2][m][8 * (c % 2) + J]; });
            return split8sx<HP>(t, bh2);
        }, cur2);
        const float c2 = opaque_s(P.sc.f2[2]);
        static_for<5>([&]<int m>() { static_for<8>([&]<int R>() { fld.v2[m][R] = o[m][R] * c2 + emb.v2[m][R]; }); });
    }
    if constexpr (L >= 3) {   // l = 3: hidden 24x3e in ONE 32-row tile; its 32 gate rows start at 0e row 336 = tile 10 row 16 (gate tile 4 upper half, tile 5 lower half)
        static_assert(f1_gate_row(3, 0) == 10 * 32 + 16, "gate rows of the l = 3 hidden block");
        f32x16 hh[7];
        const float c1 = opaque_s(P.sc.f1[3]);
        static_for<7>([&]<int m>() { static_for<16>([&]<int R>() { hh[m][R] = 0.0f; }); });
        const SR cur3 = rf1;
        rf1 = pf.template operator()<2>(O.A_f2[3], O.A_f2_l[3], 2, 0);
        gemm_sh.template operator()<7, 1>(O.A_f1[3], O.A_f1_l[3], 1, 0, hh, [&]<int m, int c>() { return parked.template operator()<3, m, c>(); }, cur3);
        static_for<7>([&]<int m>() { static_for<16>([&]<int R>() {
            if constexpr (R < 8) hh[m][R] *= gt[4][R + 8] * c1; else hh[m][R] *= gt[5][R - 8] * c1;
        }); });
        f32x16 o[7] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}};
        gemm_sh.template operator()<7, 2>(O.A_f2[3], O.A_f2_l[3], 2, 0, o, [&]<int m, int c>() {
            float t[8];
            static_for<8>([&]<int J>() { t[J] = hh[m][8 * c + J]; });
            return split8sx<HP>(t, bh3);
        }, rf1);
        const float c2 = opaque_s(P.sc.f2[3]);
        static_for<7>([&]<int m>() { static_for<8>([&]<int R>() { fld.v3[m][R] = o[m][R] * c2 + emb.v3[m][R]; }); });
    }
    sched_fence();
    // (keep this block: besides serving the stage tests it separates the FFN tail from the score stage -- ROCm 7.2 hipcc was
    //  observed to produce a wrong lmax = 2 score stage when the two end up in one basic block)
    if (P.dbg_field != nullptr && valid) {
        auto dump = [&](float* base, const Feat<L>& f) {
            float* o = base + (size_t)n * D + hi * 4;
            static_for<2>([&]<int T>() { static_for<4>([&]<int g>() {
                st4(o + T * 32 + 8 * g, f32x4{f.s[T][4 * g], f.s[T][4 * g + 1], f.s[T][4 * g + 2], f.s[T][4 * g + 3]}); }); });
            if constexpr (L >= 1) static_for<3>([&]<int m>() { static_for<4>([&]<int g>() {
                st4(o + blk_off(1) + m * 32 + 8 * g, f32x4{f.v1[m][4 * g], f.v1[m][4 * g + 1], f.v1[m][4 * g + 2], f.v1[m][4 * g + 3]}); }); });
            if constexpr (L >= 2) static_for<5>([&]<int m>() { static_for<2>([&]<int g>() {
                st4(o + blk_off(2) + m * 16 + 8 * g, f32x4{f.v2[m][4 * g], f.v2[m][4 * g + 1], f.v2[m][4 * g + 2], f.v2[m][4 * g + 3]}); }); });
            if constexpr (L >= 3) static_for<7>([&]<int m>() { static_for<2>([&]<int g>() {
                st4(o + blk_off(3) + m * 16 + 8 * g, f32x4{f.v3[m][4 * g], f.v3[m][4 * g + 1], f.v3[m][4 * g + 2], f.v3[m][4 * g + 3]}); }); });
        };
        dump(P.dbg_emb, emb);
        dump(P.dbg_field, fld);
    }

    if constexpr (UN) {
        if (valid) feat_store_ref<L>(fld, P.feat_out + (size_t)n * D, hi);
        return;
    }
    // ---- score tensor products ------------------------------------------------------------------------------------------------
    const Buf qfb = make_buf(P.qf, P.qf_bytes);
    const Buf pb = make_buf(P.pose, P.pose_bytes);
    float qraw[4], D1[9], D2[25], D3[L >= 3 ? 49 : 1];
    {
        const int pv = pose * (pose_rec<L>() * 4);
        const f32x4 t = bld4(pb, pv, 0);
        qraw[0] = t[0]; qraw[1] = t[1]; qraw[2] = t[2]; qraw[3] = t[3];
        static_for<3>([&]<int Q>() {
            const f32x4 d = bld4(pb, pv, (4 + 4 * Q) * 4);
            static_for<4>([&]<int J>() { if constexpr (4 * Q + J < 9) D1[4 * Q + J] = d[J]; });
        });
        if constexpr (L >= 2) static_for<7>([&]<int Q>() {
            const f32x4 d = bld4(pb, pv, (16 + 4 * Q) * 4);
            static_for<4>([&]<int J>() { if constexpr (4 * Q + J < 25) D2[4 * Q + J] = d[J]; });
        });
        if constexpr (L >= 3) static_for<13>([&]<int Q>() {
            const f32x4 d = bld4(pb, pv, (48 + 4 * Q) * 4);
            static_for<4>([&]<int J>() { if constexpr (4 * Q + J < 49) D3[4 * Q + J] = d[J]; });
        });
    }
    const int qv0 = q * (D * 4) + hi * 16, qv1 = q * (D * 4) + hi * 48, qv2 = q * (D * 4) + hi * 80, qv3 = q * (D * 4) + hi * 112;
    if constexpr (EBM) {
        // ---- EbmScoreModelHead.compute_energy (score_head_ebm.py:171-172): |field - D(q) f_query|^2 / dim, weighted by w_q ----
        float esum = 0.0f;
        static_for<L + 1>([&]<int l>() {
            constexpr int d = 2 * l + 1;
            static_for<mul_of(l) / 8>([&]<int gu>() {
                float xr[4 * d];
                const int qv = l == 0 ? qv0 : (l == 1 ? qv1 : (l == 2 ? qv2 : qv3));
                static_for<d>([&]<int Q>() {
                    const f32x4 t = bld4(qfb, qv, (blk_off(l) + gu * 8 * d + 4 * Q) * 4);
                    xr[4 * Q] = t[0]; xr[4 * Q + 1] = t[1]; xr[4 * Q + 2] = t[2]; xr[4 * Q + 3] = t[3];
                });
                static_for<4>([&]<int j>() {
                    static_for<d>([&]<int I>() {
                        float g, f;
                        if constexpr (l == 0) { g = xr[j]; f = fld.s[gu / 4][4 * (gu % 4) + j]; }
                        else if constexpr (l == 1) {
                            g = D1[3 * I] * xr[3 * j] + D1[3 * I + 1] * xr[3 * j + 1] + D1[3 * I + 2] * xr[3 * j + 2];
                            f = fld.v1[I][4 * gu + j];
                        } else if constexpr (l == 2) {
                            g = D2[5 * I] * xr[5 * j] + D2[5 * I + 1] * xr[5 * j + 1] + D2[5 * I + 2] * xr[5 * j + 2] +
                                D2[5 * I + 3] * xr[5 * j + 3] + D2[5 * I + 4] * xr[5 * j + 4];
                            f = fld.v2[I][4 * gu + j];
                        } else {
                            g = 0.0f;
                            static_for<7>([&]<int J>() { g += D3[7 * I + J] * xr[7 * j + J]; });
                            f = fld.v3[I][4 * gu + j];
                        }
                        esum += (f - g) * (f - g);
                    });
                });
            });
        });
        esum += xor32(esum);
        if (valid && hi == 0) {
            float* out = P.node_out + (size_t)n * 8;
            st4(out, f32x4{P.qw[q] * (esum * (1.0f / true_feat_dim<L>())), 0.0f, 0.0f, 0.0f});      // (the padded channels are 0 on both sides)
            st4(out + 4, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
        }
        return;
    }
    float res[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};      // per TP: mean over the 32 gated 1e channels
    park(fld, opaque_s(P.sc.bf));            // the field as B operands, shared by every path of both tensor products
    static_for<2>([&]<int tp>() {
        if (tp_only >= 0 && tp_only != tp) return;      // (wave-uniform: the other product is the partner wave's, NodeParams::split)
#ifndef DEDF_NODE_SPD
#define DEDF_NODE_SPD 2      // operand prefetch depth of the score tensor products' first-stage GEMMs (experiments: 3, 4)
#endif
        // first operands of the walk's first product (stp_step): in flight under the accumulator setup and the first rotation of query rows
        SharedRing<DEDF_NODE_SPD> nxt{};
        if constexpr (DEDF_NODE_CARRY) {
            constexpr StpStep s0 = stp_step<L>(0);
            constexpr int NCK0 = stp_path<L>(s0.p).mul2 / 16;
            nxt = shared_prefetch<NCK0, DEDF_NODE_SPD, HP>(wv, O.A_s[tp][s0.p], O.A_s_l[tp][s0.p], NCK0, s0.To);
        }
        f32x16 gacc = node_ldrows(rows, hi, tp == 0 ? NR::b_sl0 : NR::b_sl1, 0);
        f32x16 vacc[3];
        static_for<3>([&]<int K>() { static_for<16>([&]<int R>() { vacc[K][R] = 0.0f; }); });
#ifndef DEDF_NODE_HOIST_ROT
#define DEDF_NODE_HOIST_ROT 1
#endif
        // The query feature rows of one degree, rotated by D^{l1}(q), are the same for every path with that input degree: the paths are walked
        // grouped by l1 and the rotation (9 / 25 / 49 multiply-adds per channel) is done once per group instead of once per path.
        static_for<(DEDF_NODE_HOIST_ROT ? L + 1 : 1)>([&]<int l1h>() {
        constexpr int dh = 2 * l1h + 1, NToh = cdiv(mul_of(l1h), 32);
        float xq[DEDF_NODE_HOIST_ROT ? NToh : 1][2][2][4][DEDF_NODE_HOIST_ROT ? dh : 1];
        if constexpr (DEDF_NODE_HOIST_ROT) {
            const int qv = l1h == 0 ? qv0 : (l1h == 1 ? qv1 : (l1h == 2 ? qv2 : qv3));
            static_for<NToh>([&]<int To>() { static_for<imin(2, (mul_of(l1h) - 32 * To) / 16)>([&]<int cc>() { static_for<2>([&]<int run>() {
                constexpr int u0 = 32 * To + 16 * cc;
                float xr[4 * dh];
                static_for<dh>([&]<int Q>() {
                    const f32x4 t = bld4(qfb, qv, (blk_off(l1h) + (u0 + 8 * run) * dh + 4 * Q) * 4);
                    xr[4 * Q] = t[0]; xr[4 * Q + 1] = t[1]; xr[4 * Q + 2] = t[2]; xr[4 * Q + 3] = t[3];
                });
                static_for<4>([&]<int j>() {
                    if constexpr (l1h == 0) xq[To][cc][run][j][0] = xr[j];
                    else if constexpr (l1h == 1) static_for<3>([&]<int I>() {
                        xq[To][cc][run][j][I] = D1[3 * I] * xr[3 * j] + D1[3 * I + 1] * xr[3 * j + 1] + D1[3 * I + 2] * xr[3 * j + 2]; });
                    else if constexpr (l1h == 2) static_for<5>([&]<int I>() {
                        xq[To][cc][run][j][I] = D2[5 * I] * xr[5 * j] + D2[5 * I + 1] * xr[5 * j + 1] + D2[5 * I + 2] * xr[5 * j + 2] +
                                                D2[5 * I + 3] * xr[5 * j + 3] + D2[5 * I + 4] * xr[5 * j + 4]; });
                    else static_for<7>([&]<int I>() {
                        float t = 0.0f;
                        static_for<7>([&]<int J>() { t += D3[7 * I + J] * xr[7 * j + J]; });
                        xq[To][cc][run][j][I] = t; });
                });
            }); }); });
        }
        static_for<stp_num_paths<L>()>([&]<int p>() {
            constexpr PathInfo pi = stp_path<L>(p);
            if constexpr (!DEDF_NODE_HOIST_ROT || pi.l1 == l1h) {
            constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3;
            constexpr int d1 = 2 * l1 + 1, d2 = 2 * l2 + 1, d3 = 2 * l3 + 1;
            constexpr int NCK = pi.mul2 / 16;             // K chunks over v
            constexpr int NCL = stp_k<L>(l3) / 16;        // chunks per row tile of the final LinearRS image
            using C = CG<l1, l2, l3>;
            const float bsc = opaque_s(P.sc.s[tp][p]) * opaque_s(P.sc.bt[tp]);      // T accumulators -> true value, times the B-operand scale of stage 2
            static_for<cdiv(pi.mul1, 32)>([&]<int To>() {
                constexpr int NC2 = imin(2, (pi.mul1 - 32 * To) / 16);      // 16-channel chunks of this row tile
                // operands of the final LinearRS for this tile's chunks, requested before the first-stage GEMM
                f32x4 slh[NC2], sll[NC2] = {};
                static_for<NC2>([&]<int cc>() {
                    constexpr int ci = stp_chunk_index<L>(p, 2 * To + cc);
                    slh[cc] = bld4(wv.w, wv.lane16, (O.A_sl[tp][l3] + ci * 256) * 4);
                    if constexpr (!HP) sll[cc] = bld4(wv.w, wv.lane16, (O.A_sl_l[tp][l3] + ci * 256) * 4);
                });
                f32x16 T[d2];
                static_for<d2>([&]<int j>() { static_for<16>([&]<int R>() { T[j][R] = 0.0f; }); });
                if constexpr (DEDF_NODE_CARRY && DEDF_NODE_HOIST_ROT) {
                    const SharedRing<DEDF_NODE_SPD> cur = nxt;
                    constexpr int si = stp_step_index<L>(p, To);
                    static_assert(si >= 0, "score TP walk");
                    if constexpr (si + 1 < stp_num_steps<L>()) {
                        constexpr StpStep sn = stp_step<L>(si + 1);
                        constexpr int NCKn = stp_path<L>(sn.p).mul2 / 16;
                        nxt = shared_prefetch<NCKn, DEDF_NODE_SPD, HP>(wv, O.A_s[tp][sn.p], O.A_s_l[tp][sn.p], NCKn, sn.To);
                    }
                    dense_shared_hp<d2, NCK, DEDF_NODE_SPD, HP>(wv, O.A_s[tp][p], O.A_s_l[tp][p], NCK, To, T, [&]<int j, int c>() { return parked.template operator()<l2, j, c>(); }, cur);
                } else
                dense_shared_hp<d2, NCK, DEDF_NODE_SPD, HP>(wv, O.A_s[tp][p], O.A_s_l[tp][p], NCK, To, T, [&]<int j, int c>() { return parked.template operator()<l2, j, c>(); });
                static_for<NC2>([&]<int cc>() {
                    constexpr int u0 = 32 * To + 16 * cc;
                    float a[d3][8];
                    static_for<2>([&]<int run>() {
                        // query feature rows u0 + 8 run + 4 hi + j (reference layout), rotated by D^{l1}(q)
                        float xr[4 * d1];
                        if constexpr (!DEDF_NODE_HOIST_ROT) {
                        const int qv = l1 == 0 ? qv0 : (l1 == 1 ? qv1 : (l1 == 2 ? qv2 : qv3));
                        static_for<d1>([&]<int Q>() {
                            const f32x4 t = bld4(qfb, qv, (blk_off(l1) + (u0 + 8 * run) * d1 + 4 * Q) * 4);
                            xr[4 * Q] = t[0]; xr[4 * Q + 1] = t[1]; xr[4 * Q + 2] = t[2]; xr[4 * Q + 3] = t[3];
                        });
                        }
                        static_for<4>([&]<int j>() {
                            float x[d1], y[d2], m[C::NM], o[d3];
                            if constexpr (DEDF_NODE_HOIST_ROT) static_for<d1>([&]<int I>() { x[I] = xq[To][cc][run][j][I]; });
                            else
                            if constexpr (l1 == 0) x[0] = xr[j];
                            else if constexpr (l1 == 1) static_for<3>([&]<int I>() {
                                x[I] = D1[3 * I] * xr[3 * j] + D1[3 * I + 1] * xr[3 * j + 1] + D1[3 * I + 2] * xr[3 * j + 2]; });
                            else if constexpr (l1 == 2) static_for<5>([&]<int I>() {
                                x[I] = D2[5 * I] * xr[5 * j] + D2[5 * I + 1] * xr[5 * j + 1] + D2[5 * I + 2] * xr[5 * j + 2] +
                                       D2[5 * I + 3] * xr[5 * j + 3] + D2[5 * I + 4] * xr[5 * j + 4]; });
                            else static_for<7>([&]<int I>() {
                                float t = 0.0f;
                                static_for<7>([&]<int J>() { t += D3[7 * I + J] * xr[7 * j + J]; });
                                x[I] = t; });
                            static_for<d2>([&]<int J>() { y[J] = T[J][8 * cc + 4 * run + j]; });
                            C::make(y, m);
                            C::apply(x, m, o);
                            static_for<d3>([&]<int K>() { a[K][4 * run + j] = o[K]; });
                        });
                    });
                    const h8 ah = __builtin_bit_cast(h8, slh[cc]), al = __builtin_bit_cast(h8, sll[cc]);
                    if constexpr (l3 == 0) {
                        const HL b = split8sx<HP>(a[0], bsc);
                        gacc = mfma_h(ah, b.hi, gacc);
                        if constexpr (!HP) { gacc = mfma_h(ah, b.lo, gacc); gacc = mfma_h(al, b.hi, gacc); }
                    } else {
                        HL b[3];
                        static_for<3>([&]<int K>() { b[K] = split8sx<HP>(a[K], bsc); });
                        static_for<3>([&]<int K>() { vacc[K] = mfma_h(ah, b[K].hi, vacc[K]); });
                        if constexpr (!HP) {
                            static_for<3>([&]<int K>() { vacc[K] = mfma_h(ah, b[K].lo, vacc[K]); });
                            static_for<3>([&]<int K>() { vacc[K] = mfma_h(al, b[K].hi, vacc[K]); });
                        }
                    }
                });
                (void)NCL;
                sched_fence();
            });
            }
        });
        });
        // Gate (sigmoid on the 32 gates) and mean over the 32 1e channels (score_head.py:196-199)
        const float cg = opaque_s(P.sc.sl[tp][0]), cvv = opaque_s(P.sc.sl[tp][1]) * (1.0f / 32);
        static_for<3>([&]<int K>() {
            float s = 0.0f;
            static_for<16>([&]<int R>() { s += vacc[K][R] * sigmoid_n(gacc[R] * cg); });
            s += xor32(s);
            res[tp][K] = s * cvv;
        });
    });

    // ---- back to the body frame, orbital term, query weight (score_head.py:201-209) ------------------------------------------
    if (valid && hi == 0) {
        auto qmul = [](const float (&a)[4], const float (&b)[4], float (&o)[4]) {     // transforms.py:113-129
            o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
            o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
            o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
            o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
        };
        const float qi[4] = {qraw[0], -qraw[1], -qraw[2], -qraw[3]};
        const float qc[4] = {qraw[0], qraw[1], qraw[2], qraw[3]};          // invert(qinv)
        float rot[2][3];
        static_for<2>([&]<int tp>() {
            const float pq[4] = {0.0f, res[tp][0], res[tp][1], res[tp][2]};
            float t1[4], t2[4];
            qmul(qi, pq, t1);
            qmul(t1, qc, t2);
            rot[tp][0] = t2[1]; rot[tp][1] = t2[2]; rot[tp][2] = t2[3];
        });
        const float x0 = P.qx[3 * q] / P.lin_mult, x1 = P.qx[3 * q + 1] / P.lin_mult, x2 = P.qx[3 * q + 2] / P.lin_mult;
        const float w = P.qw[q];
        const float o0 = x1 * rot[0][2] - x2 * rot[0][1];
        const float o1 = x2 * rot[0][0] - x0 * rot[0][2];
        const float o2 = x0 * rot[0][1] - x1 * rot[0][0];
        float* out = P.node_out + (size_t)n * 8;
        if (tp_only == 1) st4(P.node_spin + (size_t)n * 4, f32x4{w * rot[1][0], w * rot[1][1], w * rot[1][2], 0.0f});
        else if (tp_only == 0) {
            st4(out, f32x4{w * rot[0][0], w * rot[0][1], w * rot[0][2], w * o0});
            st4(out + 4, f32x4{w * o1, w * o2, 0.0f, 0.0f});
        } else {
        st4(out, f32x4{w * rot[0][0], w * rot[0][1], w * rot[0][2], w * o0 + w * rot[1][0]});
        st4(out + 4, f32x4{w * o1 + w * rot[1][1], w * o2 + w * rot[1][2], 0.0f, 0.0f});
        }
    }
}

}  // namespace dedf
