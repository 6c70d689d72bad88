// Host side: canonical parameter schema (mirror of diffusion_edf_amd/params.py::param_spec, i.e. the reference
// state_dict below `score_head.`) and the packers that turn e3nn-layout weights into the lane-ordered A operands the
// kernels consume (dedf_layout.h / dedf_net.h).
#pragma once
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dedf.h"
#include "dedf_net.h"

namespace dedf {

struct ParamEntry { std::string name; size_t numel; size_t offset; };

struct ParamSpec {
    std::vector<ParamEntry> entries;
    std::map<std::string, size_t> index;
    size_t total = 0;
    void add(const std::string& n, size_t numel) {
        index[n] = entries.size();
        entries.push_back({n, numel, total});
        total += numel;
    }
    const float* get(const float* blob, const std::string& n) const {
        auto it = index.find(n);
        if (it == index.end()) throw std::runtime_error("missing parameter " + n);
        return blob + entries[it->second].offset;
    }
};

// Irreps bookkeeping at run time, for the TRUE shapes of the reference (64x0e+32x1e+16x2e+8x3e: the C ABI's parameter blob) and for the
// shapes the kernels run (8x3e zero-padded to 16x3e, FFN hidden 24x3e to 32: dedf_net.h::mul_of / hid_of).  Same path construction as
// dedf_net.h::dtp_path / stp_path (reference equiformer/tensor_product_rescale.py:352-382).
struct IrrepsRT {
    int L = 0;
    int mul[4] = {0, 0, 0, 0}, hid[4] = {0, 0, 0, 0};
    std::vector<PathInfo> dtp, stp;
    int dtp_k[4] = {0, 0, 0, 0}, stp_k[2] = {0, 0}, dtp_wn = 0, stp_wn = 0;
    IrrepsRT(int lmax, bool kernel_shapes) : L(lmax) {
        for (int l = 0; l <= L; ++l) { mul[l] = kernel_shapes ? mul_of(l) : true_mul(l); hid[l] = kernel_shapes ? hid_of(l) : true_hid(l); }
        int kc[4] = {0, 0, 0, 0};
        for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
            for (int l3 = iabs(l1 - l2); l3 <= imin(L, l1 + l2); ++l3) {
                dtp.push_back(PathInfo{l1, l2, l3, mul[l1], 1, dtp_wn, kc[l3]});
                dtp_wn += mul[l1]; kc[l3] += mul[l1];
            }
        for (int l = 0; l <= L; ++l) dtp_k[l] = kc[l];
        int ks[2] = {0, 0};
        for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
            for (int l3 = iabs(l1 - l2); l3 <= imin(1, l1 + l2); ++l3) {
                stp.push_back(PathInfo{l1, l2, l3, mul[l1], mul[l2], stp_wn, ks[l3]});
                stp_wn += mul[l1] * mul[l2]; ks[l3] += mul[l1];
            }
        stp_k[0] = ks[0]; stp_k[1] = ks[1];
    }
    int sum_mul() const { int d = 0; for (int l = 0; l <= L; ++l) d += mul[l]; return d; }
    int gates() const { int g = 0; for (int l = 1; l <= L; ++l) g += mul[l]; return g; }
    int lin0_rows() const { return mul[0] + gates(); }
    int f1_rows0() const { int r = 0; for (int l = 0; l <= L; ++l) r += hid[l]; return r; }
    size_t sq() const { size_t q = 0; for (int l = 0; l <= L; ++l) q += (size_t)mul[l] * mul[l]; return q; }
};

// the attention block shared by the score head and the UNet layers (state_dict names below `ga` / `blk`)
inline void spec_block(ParamSpec& S, const IrrepsRT& I, const dedf_config& c, const std::string& blk, const std::string& post_norm) {
    const std::string ga = blk + ".ga", rad = ga + ".sep_act.dtp_rad.";
    const int L = I.L;
    const int ch[4] = {c.fc_neurons[0], c.fc_neurons[1], c.fc_neurons[2], I.dtp_wn};
    S.add(rad + "net.0.weight", (size_t)ch[1] * ch[0]); S.add(rad + "net.0.bias", ch[1]);
    S.add(rad + "net.1.weight", ch[1]); S.add(rad + "net.1.bias", ch[1]);
    S.add(rad + "net.3.weight", (size_t)ch[2] * ch[1]); S.add(rad + "net.3.bias", ch[2]);
    S.add(rad + "net.4.weight", ch[2]); S.add(rad + "net.4.bias", ch[2]);
    S.add(rad + "net.6.weight", (size_t)ch[3] * ch[2]);
    S.add(rad + "offset", ch[3]);
    size_t lin_n = 0, val_n = 0;
    for (int l = 0; l <= L; ++l) {
        lin_n += (size_t)I.dtp_k[l] * (l == 0 ? I.lin0_rows() : I.mul[l]);
        val_n += (size_t)I.dtp_k[l] * I.mul[l];
    }
    S.add(ga + ".sep_act.lin.tp.weight", lin_n); S.add(ga + ".sep_act.lin.bias.0", I.lin0_rows());
    S.add(ga + ".sep_alpha.tp.weight", (size_t)I.dtp_k[0] * I.mul[0]); S.add(ga + ".sep_alpha.bias.0", I.mul[0]);
    S.add(ga + ".sep_value.dtp.tp.weight", I.dtp_wn);
    S.add(ga + ".sep_value.lin.tp.weight", val_n); S.add(ga + ".sep_value.lin.bias.0", I.mul[0]);
    S.add(ga + ".alpha_dot", I.mul[0]);
    S.add(ga + ".proj.tp.weight", I.sq()); S.add(ga + ".proj.bias.0", I.mul[0]);
    S.add(post_norm + ".affine_weight", I.sum_mul()); S.add(post_norm + ".affine_bias", I.mul[0]);
    size_t f1 = 0, f2 = 0;
    for (int l = 0; l <= L; ++l) {
        f1 += (size_t)I.mul[l] * (l == 0 ? I.f1_rows0() : I.hid[l]);
        f2 += (size_t)I.hid[l] * I.mul[l];
    }
    S.add(blk + ".ffn.fctp_1.tp.weight", f1); S.add(blk + ".ffn.fctp_1.bias.0", I.f1_rows0());
    S.add(blk + ".ffn.fctp_2.tp.weight", f2); S.add(blk + ".ffn.fctp_2.bias.0", I.mul[0]);
}

// one UNet layer: state_dict names of the ModuleDict {'radial', 'gnn'} (reference unet_feature_extractor.py:141-156; block.py:62-139,
// graph_attention.py:11-82).  norm_1_src / norm_1_dst exist in the reference's state dict and are dead (block.py:149-153).
// (UNet layers take their parameters in the KERNEL shapes: diffusion_edf_amd/unet_pad.py builds the zero-padded tensors)
inline ParamSpec build_spec_unet_layer(const IrrepsRT& I, const dedf_config& c) {
    ParamSpec S;
    const int nb = c.fc_neurons[0];
    S.add("radial.mean", nb); S.add("radial.std_logit", nb); S.add("radial.weight_logit", nb);
    const std::string g = "gnn";
    S.add(g + ".norm_1_src.affine_weight", I.sum_mul()); S.add(g + ".norm_1_src.affine_bias", I.mul[0]);
    S.add(g + ".linear_src.tp.weight", I.sq());
    S.add(g + ".norm_1_dst.affine_weight", I.sum_mul()); S.add(g + ".norm_1_dst.affine_bias", I.mul[0]);
    S.add(g + ".linear_dst.tp.weight", I.sq()); S.add(g + ".linear_dst.bias.0", I.mul[0]);
    spec_block(S, I, c, g, g + ".norm_2");
    return S;
}

// canonical parameter list of a handle for irreps `I` (I = true shapes: the C ABI's blob; I = kernel shapes: what the packers read)
inline ParamSpec build_spec(const IrrepsRT& I, const dedf_config& c) {
    if (c.unet_layer) return build_spec_unet_layer(I, c);
    ParamSpec S;
    const int* te = c.time_emb_mlp;
    for (int n = 0; n < c.n_scales; ++n) {
        const std::string p = "time_mlps_multiscale." + std::to_string(n) + ".";
        S.add(p + "0.weight", (size_t)te[1] * te[0]); S.add(p + "0.bias", te[1]);
        S.add(p + "2.weight", (size_t)te[2] * te[1]); S.add(p + "2.bias", te[2]);
    }
    if (c.query_time_encoding) {      // score_head.py:64-70
        S.add("query_time_mlp.0.weight", (size_t)te[1] * te[0]); S.add("query_time_mlp.0.bias", te[1]);
        S.add("query_time_mlp.2.weight", (size_t)te[2] * te[1]); S.add("query_time_mlp.2.bias", te[2]);
    }
    const std::string ktf = "key_tensor_field";
    const int F0 = c.fc_neurons[0], dimL = c.length_emb_dim;
    for (int n = 0; n < c.n_scales; ++n) {
        if (c.radii[n] > 0) {
            const std::string pm = ktf + ".graph_parsers." + std::to_string(n) + ".length_enc.param_module.";
            S.add(pm + "std_logit", dimL); S.add(pm + "weight_logit", dimL); S.add(pm + "mean", dimL);
        }
        const std::string pl = ktf + ".edge_scalars_pre_linears." + std::to_string(n) + ".0.";
        S.add(pl + "weight", (size_t)F0 * F0); S.add(pl + "bias", F0);
    }
    const std::string blk = ktf + ".gnn_block_init";
    S.add(blk + ".prenorm_src.affine_weight", I.sum_mul()); S.add(blk + ".prenorm_src.affine_bias", I.mul[0]);
    S.add(blk + ".linear_src.tp.weight", I.sq());
    if (!c.query_time_encoding) S.add(blk + ".linear_src.bias.0", I.mul[0]);
    else {      // use_dst_feature = True with irreps_dst = te[2] x 0e (gnn_block.py:109-130; score_head.py:52, 81-83)
        S.add(blk + ".skip_1.skip.tp.weight", (size_t)te[2] * I.mul[0]); S.add(blk + ".skip_1.skip.bias.0", I.mul[0]);
        S.add(blk + ".prenorm_dst.affine_weight", te[2]); S.add(blk + ".prenorm_dst.affine_bias", te[2]);
        S.add(blk + ".linear_dst.tp.weight", (size_t)te[2] * I.mul[0]); S.add(blk + ".linear_dst.bias.0", I.mul[0]);
    }
    spec_block(S, I, c, blk, blk + ".post_norm");
    if (!c.ebm) for (const char* nm : {"lin_vel_tp", "ang_vel_tp"}) {
        const std::string p = std::string(nm);
        S.add(p + ".dtp.tp.weight", I.stp_wn);
        S.add(p + ".lin.tp.weight", (size_t)I.stp_k[0] * (1 + I.mul[1]) + (size_t)I.stp_k[1] * I.mul[1]);
        S.add(p + ".lin.bias.0", 1 + I.mul[1]);
    }
    return S;
}

// True-shape parameter blob -> kernel-shape blob (only lmax 3 differs: 8x3e -> 16x3e, FFN hidden 24x3e -> 32x3e).  Every tensor that
// indexes a 3e channel is re-laid with the true channels at dedf_net.h::pad_pos and zeros elsewhere (C++ twin of
// diffusion_edf_amd/unet_pad.py::expand_layer_params, which does the same for the narrow UNet levels).
inline std::vector<float> pad_params(const dedf_config& c, const IrrepsRT& T, const ParamSpec& St, const IrrepsRT& K, const ParamSpec& Sk, const float* Bt) {
    std::vector<float> out(Sk.total, 0.0f);
    const int L = T.L;
    auto src = [&](const std::string& n) { return St.get(Bt, n); };
    auto dst = [&](const std::string& n) { return out.data() + Sk.entries[Sk.index.at(n)].offset; };
    auto numel = [&](const ParamSpec& S, const std::string& n) { return S.entries[S.index.at(n)].numel; };
    // index maps true -> kernel
    auto pos = [&](int l, int ch) { return pad_pos(l, ch); };
    std::vector<int> wmap(T.dtp_wn);                               // flat depth-wise-TP weight index (creation order)
    std::vector<std::vector<int>> kmap(L + 1);                     // sorted DTP channel inside the l3 block
    for (int l = 0; l <= L; ++l) kmap[l].assign(T.dtp_k[l], -1);
    for (size_t p = 0; p < T.dtp.size(); ++p)
        for (int u = 0; u < T.dtp[p].mul1; ++u) {
            wmap[T.dtp[p].wstart + u] = K.dtp[p].wstart + pos(T.dtp[p].l1, u);
            kmap[T.dtp[p].l3][T.dtp[p].kofs + u] = K.dtp[p].kofs + pos(T.dtp[p].l1, u);
        }
    std::vector<int> lin0(T.lin0_rows()), f1r(T.f1_rows0());       // 0e row spaces of sep_act.lin (scalars | gates) and fctp_1 (hidden scalars | gates)
    {
        for (int i = 0; i < T.mul[0]; ++i) lin0[i] = i;
        int ot = T.mul[0], ok = K.mul[0];
        for (int l = 1; l <= L; ++l) { for (int i = 0; i < T.mul[l]; ++i) lin0[ot + i] = ok + pos(l, i); ot += T.mul[l]; ok += K.mul[l]; }
        ot = 0; ok = 0;
        for (int l = 0; l <= L; ++l) { for (int i = 0; i < T.hid[l]; ++i) f1r[ot + i] = ok + i; ot += T.hid[l]; ok += K.hid[l]; }
    }
    std::vector<std::vector<int>> smap(2);                         // sorted score-TP channel inside the l3 block
    smap[0].assign(T.stp_k[0], -1); smap[1].assign(T.stp_k[1], -1);
    for (size_t p = 0; p < T.stp.size(); ++p)
        for (int u = 0; u < T.stp[p].mul1; ++u) smap[T.stp[p].l3][T.stp[p].kofs + u] = K.stp[p].kofs + pos(T.stp[p].l1, u);
    // helpers: vector with an index map; matrix [rows][cols] with row / column maps
    auto vec = [&](const std::string& n, const std::vector<int>& m) { const float* s_ = src(n); float* d_ = dst(n); for (size_t i = 0; i < m.size(); ++i) d_[m[i]] = s_[i]; };
    auto mat = [&](const float* s_, float* d_, int R, int Cn, int Ck, auto rmap, auto cmap) {
        for (int r = 0; r < R; ++r) for (int q = 0; q < Cn; ++q) d_[(size_t)rmap(r) * Ck + cmap(q)] = s_[(size_t)r * Cn + q];
    };
    auto ident = [](int i) { return i; };
    auto per_l_vec = [&](const std::string& n) {          // [sum_l mul_l]
        const float* s_ = src(n); float* d_ = dst(n);
        int ot = 0, ok = 0;
        for (int l = 0; l <= L; ++l) { for (int i = 0; i < T.mul[l]; ++i) d_[ok + pos(l, i)] = s_[ot + i]; ot += T.mul[l]; ok += K.mul[l]; }
    };
    auto per_l_sq = [&](const std::string& n) {           // per degree [mul in][mul out]
        const float* s_ = src(n); float* d_ = dst(n);
        size_t ot = 0, ok = 0;
        for (int l = 0; l <= L; ++l) {
            mat(s_ + ot, d_ + ok, T.mul[l], T.mul[l], K.mul[l], [&](int r) { return pos(l, r); }, [&](int q) { return pos(l, q); });
            ot += (size_t)T.mul[l] * T.mul[l]; ok += (size_t)K.mul[l] * K.mul[l];
        }
    };
    auto block = [&](const std::string& blk, const std::string& post_norm) {
        const std::string ga = blk + ".ga", rad = ga + ".sep_act.dtp_rad.";
        const int H2 = c.fc_neurons[2];
        mat(src(rad + "net.6.weight"), dst(rad + "net.6.weight"), T.dtp_wn, H2, H2, [&](int r) { return wmap[r]; }, ident);
        vec(rad + "offset", wmap);
        vec(ga + ".sep_value.dtp.tp.weight", wmap);
        {   // sep_act.lin / sep_value.lin: per l3 [k sorted DTP channel][out row]
            const float *ls = src(ga + ".sep_act.lin.tp.weight"), *vs = src(ga + ".sep_value.lin.tp.weight");
            float *ld = dst(ga + ".sep_act.lin.tp.weight"), *vd = dst(ga + ".sep_value.lin.tp.weight");
            size_t lt = 0, lk = 0, vt = 0, vk = 0;
            for (int l = 0; l <= L; ++l) {
                const int Ot = l == 0 ? T.lin0_rows() : T.mul[l], Ok = l == 0 ? K.lin0_rows() : K.mul[l];
                mat(ls + lt, ld + lk, T.dtp_k[l], Ot, Ok, [&](int r) { return kmap[l][r]; }, [&](int q) { return l == 0 ? lin0[q] : pos(l, q); });
                mat(vs + vt, vd + vk, T.dtp_k[l], T.mul[l], K.mul[l], [&](int r) { return kmap[l][r]; }, [&](int q) { return pos(l, q); });
                lt += (size_t)T.dtp_k[l] * Ot; lk += (size_t)K.dtp_k[l] * Ok; vt += (size_t)T.dtp_k[l] * T.mul[l]; vk += (size_t)K.dtp_k[l] * K.mul[l];
            }
        }
        vec(ga + ".sep_act.lin.bias.0", lin0);
        mat(src(ga + ".sep_alpha.tp.weight"), dst(ga + ".sep_alpha.tp.weight"), T.dtp_k[0], T.mul[0], K.mul[0], [&](int r) { return kmap[0][r]; }, ident);
        per_l_sq(ga + ".proj.tp.weight");
        per_l_vec(post_norm + ".affine_weight");
        {   // FFN: fctp_1 per degree [mul in][rows out], fctp_2 per degree [hidden in][mul out]
            const float *s1 = src(blk + ".ffn.fctp_1.tp.weight"), *s2 = src(blk + ".ffn.fctp_2.tp.weight");
            float *d1 = dst(blk + ".ffn.fctp_1.tp.weight"), *d2 = dst(blk + ".ffn.fctp_2.tp.weight");
            size_t t1 = 0, k1 = 0, t2 = 0, k2 = 0;
            for (int l = 0; l <= L; ++l) {
                const int Ot = l == 0 ? T.f1_rows0() : T.hid[l], Ok = l == 0 ? K.f1_rows0() : K.hid[l];
                mat(s1 + t1, d1 + k1, T.mul[l], Ot, Ok, [&](int r) { return pos(l, r); }, [&](int q) { return l == 0 ? f1r[q] : q; });
                mat(s2 + t2, d2 + k2, T.hid[l], T.mul[l], K.mul[l], ident, [&](int q) { return pos(l, q); });
                t1 += (size_t)T.mul[l] * Ot; k1 += (size_t)K.mul[l] * Ok; t2 += (size_t)T.hid[l] * T.mul[l]; k2 += (size_t)K.hid[l] * K.mul[l];
            }
        }
        vec(blk + ".ffn.fctp_1.bias.0", f1r);
    };
    // 1. everything whose shape does not involve a 3e channel: straight copy
    for (const ParamEntry& e : St.entries) {
        auto it = Sk.index.find(e.name);
        if (it == Sk.index.end()) throw std::runtime_error("pad_params: " + e.name + " missing in the kernel schema");
        if (Sk.entries[it->second].numel == e.numel) std::copy(Bt + e.offset, Bt + e.offset + e.numel, out.data() + Sk.entries[it->second].offset);
    }
    if (L < 3 || c.unet_layer) return out;      // (UNet-layer handles already speak the kernel shapes)
    // 2. the tensors that do
    const std::string blk = "key_tensor_field.gnn_block_init";
    per_l_vec(blk + ".prenorm_src.affine_weight");
    per_l_sq(blk + ".linear_src.tp.weight");
    block(blk, blk + ".post_norm");
    if (!c.ebm) for (const char* nm : {"lin_vel_tp", "ang_vel_tp"}) {
        const std::string p = std::string(nm);
        const float* s_ = src(p + ".dtp.tp.weight"); float* d_ = dst(p + ".dtp.tp.weight");
        for (size_t q = 0; q < T.stp.size(); ++q)
            mat(s_ + T.stp[q].wstart, d_ + K.stp[q].wstart, T.stp[q].mul1, T.stp[q].mul2, K.stp[q].mul2,
                [&](int r) { return pos(T.stp[q].l1, r); }, [&](int v) { return pos(T.stp[q].l2, v); });
        const float* ls = src(p + ".lin.tp.weight"); float* ld = dst(p + ".lin.tp.weight");
        const int O0 = 1 + T.mul[1];
        mat(ls, ld, T.stp_k[0], O0, O0, [&](int r) { return smap[0][r]; }, ident);
        mat(ls + (size_t)T.stp_k[0] * O0, ld + (size_t)K.stp_k[0] * O0, T.stp_k[1], T.mul[1], T.mul[1], [&](int r) { return smap[1][r]; }, ident);
    }
    (void)numel;
    return out;
}

// ---- growing image of packed weights: every block is 16-byte aligned, offsets are in floats --------------------------------
struct Image {
    std::vector<float> data;
    int push(const std::vector<float>& v) {
        while (data.size() % 4) data.push_back(0.0f);
        const int off = (int)data.size();
        data.insert(data.end(), v.begin(), v.end());
        return off;
    }
};

struct EdgeOffsets {
    int o_enc, o_A_pre, o_A_r1, o_b_r1, o_g_r1, o_be_r1, o_A_r2, o_b_r2, o_g_r2, o_be_r2, o_A_r3, o_off_r3;
    int o_A_pre_l, o_A_r1_l, o_A_r2_l, o_A_r3_l;      // residual (lo) images of the split-fp16 radial-MLP layers
    int o_S_lin, o_b_r0, o_S_val, o_b_val0, o_alpha_dot;     // o_S_*: split-fp16 A-operand streams (pack_dtp_stream)
    // power-of-two operand scaling of the split-fp16 GEMMs (keeps the lo halves out of the fp16 subnormal range, see pow2_scale):
    float w_unscale;        // layer-3 output (per-edge TP weights) carries 2^s3; this is 2^-s3
    float u_scale;          // the gated features are parked as 2^su * u
    float c_lin[4], c_val[4];   // accumulator -> true value of the lin / value GEMMs, per output degree
    float est_value;            // typical magnitude of the attention value (= of the aggregate z the node kernel reads), estimated from the weights
    int s3, su;                 // the two activation-side exponents chosen for this handle (tests: DEDF_SCALE_DEBUG)
    float msg_scale;            // power of two the source message is stored with (k_src_message): s3 = exponent of the weight image + log2(msg_scale)
};
struct NodeOffsets {
    int o_A_proj[4], o_b_proj0, o_ln_w[4], o_ln_b0, o_A_f1[4], o_b_f1, o_A_f2[4], o_b_f2;
    int o_A_s[2][16], o_A_sl[2][2], o_b_sl[2];
    int o_A_proj_l[4], o_A_f1_l[4], o_A_f2_l[4], o_A_s_l[2][16], o_A_sl_l[2][2];     // residual (lo) images
    NodeScales sc;
    bool layout_ok;         // the pushes reproduced dedf_net.h::kNodeLayout (checked in pack_node, enforced in dedf_create)
};

// Activation-side (B) operands are data: their size is not known when the weights are packed.  A value keeps its full 22 bits
// while it is >= 2^-3 (below that the fp16 residual is subnormal: absolute error 2^-25) and overflows at 65 504, so the typical
// magnitude is placed near the geometric middle of that window (~2^5 ... 2^7: a factor ~1000 of headroom either way) instead of
// the ~2^9 the weight images use.  kActHeadroomBits = how far below the weight-side target 2^8..2^9.
constexpr int kActHeadroomBits = 3;
// Round 5: the activation-side exponents are no longer constants.  Every activation that becomes a B operand sits behind a LayerNorm or is a
// (bounded) function of one, times weights: its typical magnitude follows from the WEIGHTS of the handle -- rms row norms propagated from the
// normalised inputs through the block (act_exponent below) --, so the exponent of every such operand is chosen at pack time to put the typical
// value at 2^kActTarget = 8: x8 000 below the fp16 maximum, and an absolute rounding error of 2^-25 (the fp16 subnormal step of the residual half)
// stays 2^-28 of the typical value.  A checkpoint whose edge linears are 100x larger than random init gets exponents 7 bits lower and the same
// relative accuracy (tests/test_gpu_parity.py::test_fp16_operand_range_scaled_weights_and_features).  What stays data is the caller's features:
// key features pass a LayerNorm first; the query features enter the score tensor products with x8 000 of room.
constexpr int kActTarget = 3;
inline int act_exponent(float typical, int lo = -40, int hi = 12) {
    if (!(typical > 0.0f) || !std::isfinite(typical)) return 0;
    const int e = kActTarget - (int)std::ceil(std::log2(typical));
    return e < lo ? lo : (e > hi ? hi : e);
}
// rms over the rows of  sqrt(sum_k W(o, k)^2)
template <class WAt> inline float rms_row_norm(int O, int K, WAt W) {
    double acc = 0.0;
    for (int o = 0; o < O; ++o) for (int k = 0; k < K; ++k) { const double w = W(o, k); acc += w * w; }
    return O > 0 ? (float)std::sqrt(acc / O) : 0.0f;
}
inline float softplusf(float x) { return x > 20.0f ? x : std::log1p(std::exp(x)); }
// exponent s such that 2^s * maxabs lands in [256, 512]: typical elements are then O(10..100), their fp16 residuals (2^-11 of
// that) stay normal fp16 numbers, and the largest element is far from the fp16 maximum.  Scaling by 2^s is exact.
inline int pow2_scale(float maxabs, int lo, int hi) {
    if (!(maxabs > 0.0f)) return lo;
    const int s = (int)std::floor(std::log2(512.0f / maxabs));
    return s < lo ? lo : (s > hi ? hi : s);
}

// Split-fp16 A-operand stream of a linear layer fed by the depth-wise TP (dedf_net.h::dtp_item): slots in the order the
// edge kernel consumes them.  Slot = 512 floats: hi image (256) | lo image (256) of one (output tile x chunk) operand,
// [lane 64][8 halves] for v_mfma_f32_32x32x16_f16 (row lane & 31, k16 = 8 (lane >> 5) + j); rows past rows[l] are zero.
//   W(l, o, k) -> weight of output row o of block l for sorted DTP channel k;  rows[l] = valid output rows
template <int L, bool S = false, class WAt>
inline std::vector<float> pack_dtp_stream(int nt0, const int* rows, WAt W) {
    std::vector<uint16_t> img((size_t)dtp_num_slots<L, S>(nt0) * 1024, 0);
    auto put = [&](size_t half_idx, float w) {
        const _Float16 hh = (_Float16)w;
        __builtin_memcpy(&img[half_idx], &hh, 2);
        return (float)hh;
    };
    size_t slot = 0;
    for (int pos = 0; pos < dtp_wn<L>() / 16; ++pos) {
        const int l3 = dtp_pos_l3<L, S>(pos);
        const int nt = l3 == 0 ? nt0 : 1;
        for (int To = 0; To < nt; ++To, ++slot)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = To * 32 + (lane & 31), k16 = 8 * (lane >> 5) + j;
                    if (o >= rows[l3]) continue;
                    const float w = W(l3, o, dtp_pos_channel<L, S>(pos, k16));
                    const size_t hi_idx = slot * 1024 + (size_t)lane * 8 + j;
                    const float h = put(hi_idx, w);
                    put(hi_idx + 512, w - h);
                }
    }
    std::vector<float> out(img.size() / 2);
    __builtin_memcpy(out.data(), img.data(), img.size() * 2);
    return out;
}

// Split-fp16 A-operand stream of the attention-value GEMMs in output-side form (dedf_net.h::make_val_walk): one slot per
// (path, output tile, K-chunk) in the order the items first use them; same slot format as pack_dtp_stream.  The K index of
// element (h = lane >> 5, j) is channel 16 c + rowmap(j, h) of the path's input block: the order in which the gated features were
// parked (8 consecutive accumulator registers of each half-wave).
//   W(p, o, u) -> weight from input channel u of path p to output row o of the l3 block
template <int L, class WAt>
inline std::vector<float> pack_val_stream(WAt W) {
    std::vector<uint16_t> img((size_t)val_num_slots<L>() * 1024, 0);
    auto put = [&](size_t half_idx, float w) {
        const _Float16 hh = (_Float16)w;
        __builtin_memcpy(&img[half_idx], &hh, 2);
        return (float)hh;
    };
    std::vector<char> done(val_num_slots<L>(), 0);
    for (int I = 0; I < val_num_items<L>(); ++I) {
        const VItem it = val_item<L>(I);
        const PathInfo pi = dtp_path<L>(it.p);
        for (int a = 0; a < it.na; ++a) {
            const int slot = it.aslot[a];
            if (done[slot]) continue;
            done[slot] = 1;
            // chunk of this accumulator: recover it from the parked slot index
            const int c = it.bq[a] - park_slot<L>(pi.l1, it.in_side ? 0 : it.comp[a], 0);      // (input-side items: bq = component 0 of their chunk)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = (it.share_b ? a : it.t) * 32 + (lane & 31), u = 16 * c + rowmap(j, lane >> 5);
                    if (o >= mul_of(pi.l3)) continue;
                    const float w = W(it.p, o, u);
                    const size_t hi_idx = (size_t)slot * 1024 + (size_t)lane * 8 + j;
                    const float h = put(hi_idx, w);
                    put(hi_idx + 512, w - h);
                }
        }
    }
    std::vector<float> out(img.size() / 2);
    __builtin_memcpy(out.data(), img.data(), img.size() * 2);
    return out;
}

// Split-fp16 A-operand stream of the attention value in the edge frame (dedf_net.h::make_sval_walk): one slot per (path, K-chunk, coefficient
// class) -- per output tile for the scalar outputs -- in the order the items use them, the class's coefficient folded in.
//   W(p, o, u) -> weight from input channel u of path p to output row o of the l3 block
template <int L, class WAt>
inline std::vector<float> pack_sval_stream(WAt W) {
    std::vector<uint16_t> img((size_t)sval_num_slots<L>() * 1024, 0);
    auto put = [&](size_t half_idx, float w) {
        const _Float16 hh = (_Float16)w;
        __builtin_memcpy(&img[half_idx], &hh, 2);
        return (float)hh;
    };
    std::vector<char> done(sval_num_slots<L>(), 0);
    for (int I = 0; I < sval_num_items<L>(); ++I) {
        const SItem it = sval_item<L>(I);
        const PathInfo pi = dtp_path<L>(it.p);
        for (int a = 0; a < it.na; ++a) {
            const int slot = it.aslot[a];
            if (done[slot]) continue;
            done[slot] = 1;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = (it.l3 == 0 ? it.acc[a] : 0) * 32 + (lane & 31), u = 16 * it.c + rowmap(j, lane >> 5);
                    if (o >= mul_of(pi.l3)) continue;
                    const float w = W(it.p, o, u) * it.coef;
                    const size_t hi_idx = (size_t)slot * 1024 + (size_t)lane * 8 + j;
                    const float h = put(hi_idx, w);
                    put(hi_idx + 512, w - h);
                }
        }
    }
    std::vector<float> out(img.size() / 2);
    __builtin_memcpy(out.data(), img.data(), img.size() * 2);
    return out;
}

// SO2: the image of the edge-aligned-frame kernels (dedf_net.h::make_dtp_walk_so2 / make_sval_walk): the last radial layer's rows and the lin /
// sep_alpha stream in that walk's order with kSo2Ref folded into the l3 >= 1 slots, the value stream per coefficient class
template <int L, bool SO2 = false>
inline void pack_edge(const dedf_config& c, const ParamSpec& S, const float* B, Image& im, EdgeOffsets& o) {
    const bool un = c.unet_layer != 0;
    const std::string ktf = "key_tensor_field", blk = un ? std::string("gnn") : ktf + ".gnn_block_init", ga = blk + ".ga", rad = ga + ".sep_act.dtp_rad.";
    // length encoders (reference radial_func.py:168-227, 291-316; UNet layer: GaussianRadialBasisLayerFiniteCutoff :231-278, whose
    // Gaussian part has the same three parameter rows -- its rescaling of the length and its soft cut-off are kernel arguments)
    {
        std::vector<float> enc((size_t)c.n_scales * 192, 0.0f);
        for (int n = 0; n < c.n_scales; ++n) {
            float* e = enc.data() + (size_t)n * 192;
            if (c.radii[n] > 0) {
                const std::string pm = un ? std::string("radial.") : ktf + ".graph_parsers." + std::to_string(n) + ".length_enc.param_module.";
                const float *sl = S.get(B, pm + "std_logit"), *wl = S.get(B, pm + "weight_logit"), *mu = S.get(B, pm + "mean");
                for (int k = 0; k < 64; ++k) {
                    const int hi = k / 32, s = k % 32;
                    const float sd = softplusf(sl[k]) + 1e-5f;
                    e[0 + hi * 32 + s] = mu[k] + 0.0f;
                    e[64 + hi * 32 + s] = 1.0f / sd;
                    // (UNet layer narrower than instantiated: the normaliser is sqrt(true num_basis), radial_func.py:259)
                    const float nbasis = (float)(un && c.unet_fc_valid[0] > 0 ? c.unet_fc_valid[0] : c.length_emb_dim);
                    e[128 + hi * 32 + s] = (1.0f / (1.0f + std::exp(-wl[k]))) * (4.0f * std::sqrt(nbasis));
                }
            } else {
                const double step = std::log(1000.0) / (32 - 1);
                for (int k = 0; k < 32; ++k) e[k] = std::exp((float)k * (float)(-step));
            }
        }
        o.o_enc = im.push(enc);
    }
    // The radial MLP (pre-linear, layers 1-3) runs on split-fp16 MFMAs: hi / lo images per layer (dedf_layout.h::pack_A_h)
    const int F0 = c.fc_neurons[0];      // 64 + time channels; 64 for the EBM head
    if (un) { o.o_A_pre = o.o_A_pre_l = 0; }
    else {   // pre-linear: length-embedding columns of the weight (multiscale_tensor_field.py:141-146); B operand element j of chunk cc
        // is the lane's embedding value 8 cc + j, i.e. k = 8 cc + j + 32 h
        std::vector<float> all_h, all_l;
        for (int n = 0; n < c.n_scales; ++n) {
            const float* W = S.get(B, ktf + ".edge_scalars_pre_linears." + std::to_string(n) + ".0.weight");
            std::vector<float> ih, il;
            pack_A_h(F0, 4, [&](int oo, int k) { return W[oo * F0 + k]; }, [](int cc, int j, int h) { return 8 * cc + j + 32 * h; }, ih, il);
            all_h.insert(all_h.end(), ih.begin(), ih.end());
            all_l.insert(all_l.end(), il.begin(), il.end());
        }
        o.o_A_pre = im.push(all_h);
        o.o_A_pre_l = im.push(all_l);
    }
    auto rows = [&](int O, const float* v) { return pack_rows(O, [&](int i) { return v[i]; }); };
    int s3 = 0;
    float est_we = 1.0f, est_msg = 0.0f;      // typical per-edge TP weight / source-message component (score heads; a UNet layer's features are not normalised: constants)
    const int H1 = c.fc_neurons[1], H2 = c.fc_neurons[2];      // hidden widths of the radial MLP: 128, 64 or 32, 32
    {
        std::vector<float> ih, il;
        const float* W = S.get(B, rad + "net.0.weight");
        // K order of layer 1: the previous layer's accumulator tiles; for a UNet layer the length embedding itself (no pre-linear):
        // element j of chunk cc is the lane's embedding value 8 cc + j, i.e. k = 8 cc + j + 32 h
        pack_A_h(H1, F0 / 16, [&](int oo, int k) { return W[oo * F0 + k]; },
                 [&](int cc, int j, int h) { return un ? 8 * cc + j + 32 * h : chain_k(F0, cc, j, h); }, ih, il);
        o.o_A_r1 = im.push(ih); o.o_A_r1_l = im.push(il);
        o.o_b_r1 = im.push(rows(H1, S.get(B, rad + "net.0.bias")));
        o.o_g_r1 = im.push(rows(H1, S.get(B, rad + "net.1.weight")));
        o.o_be_r1 = im.push(rows(H1, S.get(B, rad + "net.1.bias")));
        const float* W2 = S.get(B, rad + "net.3.weight");
        pack_A_h(H2, H1 / 16, [&](int oo, int k) { return W2[oo * H1 + k]; }, [&](int cc, int j, int h) { return chain_k(H1, cc, j, h); }, ih, il);
        o.o_A_r2 = im.push(ih); o.o_A_r2_l = im.push(il);
        o.o_b_r2 = im.push(rows(H2, S.get(B, rad + "net.3.bias")));
        o.o_g_r2 = im.push(rows(H2, S.get(B, rad + "net.4.weight")));
        o.o_be_r2 = im.push(rows(H2, S.get(B, rad + "net.4.bias")));
        const float* W3 = S.get(B, rad + "net.6.weight");
        // rows (= per-edge TP weights) in the kernel's walk order: chunks grouped by output degree (dedf_net.h::dtp_walk_row)
        const float* off3 = S.get(B, rad + "offset");
        // the layer's output w (and with it the B operands w * CG of the lin / sep_alpha GEMMs) is produced as 2^s3 * w, s3 <= 8:
        // the largest layer-3 weight lands in [32, 64] (activation-side operand, see kActHeadroomBits)
        float m3 = 0.0f;
        for (int i = 0; i < dtp_wn<L>() * H2; ++i) m3 = std::fmax(m3, std::fabs(W3[i]));
        for (int i = 0; i < dtp_wn<L>(); ++i) m3 = std::fmax(m3, std::fabs(off3[i]));
        s3 = pow2_scale(m3, -16, 8 + kActHeadroomBits) - kActHeadroomBits;
        int s3img = s3;
        o.msg_scale = 1.0f;
        if (!un) {
            // typical per-edge TP weight x typical source-message component (both follow from weights behind a LayerNorm) -> exponent of the
            // B operands  w . x'  of the lin / sep_alpha GEMMs; never above what keeps the weight image itself in range
            const float* g2 = S.get(B, rad + "net.4.weight"); const float* be2 = S.get(B, rad + "net.4.bias");
            double q2 = 0.0;
            for (int hh = 0; hh < H2; ++hh) q2 += (double)g2[hh] * g2[hh] + (double)be2[hh] * be2[hh];
            q2 /= H2;
            double acc = 0.0;
            for (int r = 0; r < dtp_wn<L>(); ++r) { double rn = 0.0; for (int k = 0; k < H2; ++k) rn += (double)W3[r * H2 + k] * W3[r * H2 + k]; acc += rn * q2 + (double)off3[r] * off3[r]; }
            est_we = (float)std::sqrt(acc / dtp_wn<L>());
            const float* ws = S.get(B, blk + ".linear_src.tp.weight"); const float* aw = S.get(B, blk + ".prenorm_src.affine_weight");
            const float* ab = S.get(B, blk + ".prenorm_src.affine_bias"); const float* bs = c.query_time_encoding ? nullptr : S.get(B, blk + ".linear_src.bias.0");
            size_t wo = 0; int ao = 0;
            for (int l = 0; l <= L; ++l) {
                const int m = mul_of(l);
                double a2 = 0.0;
                for (int w = 0; w < m; ++w) for (int u = 0; u < m; ++u) { const double in2 = (double)aw[ao + u] * aw[ao + u] + (l == 0 ? (double)ab[u] * ab[u] : 0.0); a2 += (double)ws[wo + (size_t)u * m + w] * ws[wo + (size_t)u * m + w] * in2; }
                if (l == 0 && bs != nullptr) for (int w = 0; w < m; ++w) a2 += (double)bs[w] * bs[w];
                if (l == 0 && c.query_time_encoding) {      // + the destination message LinearRS(LayerNorm(time scalars)) that joins the 0e block (gnn_block.py:172-180)
                    const int tq = c.time_emb_mlp[2];
                    const float* wd = S.get(B, blk + ".linear_dst.tp.weight"); const float* bd = S.get(B, blk + ".linear_dst.bias.0");
                    const float* dw = S.get(B, blk + ".prenorm_dst.affine_weight"); const float* db = S.get(B, blk + ".prenorm_dst.affine_bias");
                    for (int w = 0; w < m; ++w) {
                        a2 += (double)bd[w] * bd[w];
                        for (int u = 0; u < tq; ++u) a2 += (double)wd[(size_t)u * m + w] * wd[(size_t)u * m + w] * ((double)dw[u] * dw[u] + (double)db[u] * db[u]);
                    }
                }
                est_msg = std::fmax(est_msg, (float)std::sqrt(a2 / m));
                wo += (size_t)m * m; ao += m;
            }
            // (the exponent of the product is split: the weight image keeps the weight-side scale -- largest entry in [256, 512] like every A operand --,
            //  the rest rides on the stored message: a 1000x larger linear_src must not push the layer-3 weights into the fp16 subnormals)
            const int eB = act_exponent(est_we * est_msg * 2.3f);
            s3img = pow2_scale(m3, -16, 20);
            o.msg_scale = std::ldexp(1.0f, eB - s3img);
            s3 = eB;
        }
        const float f3 = std::ldexp(1.0f, s3img);
        pack_A_h(dtp_wn<L>(), H2 / 16, [&](int oo, int k) { return W3[dtp_walk_row<L, SO2>(oo) * H2 + k] * f3; }, [&](int cc, int j, int h) { return chain_k(H2, cc, j, h); }, ih, il);
        o.o_A_r3 = im.push(ih); o.o_A_r3_l = im.push(il);
        o.o_off_r3 = im.push(pack_rows(dtp_wn<L>(), [&](int i) { return off3[dtp_walk_row<L, SO2>(i)] * f3; }));
        o.w_unscale = std::ldexp(1.0f, -s3img);
    }
    {   // sep_act.lin (+ sep_alpha on the l3 = 0 K-steps) and sep_value.lin (shared DTP weights folded in)
        const float* lw = S.get(B, ga + ".sep_act.lin.tp.weight");
        const float* aw = S.get(B, ga + ".sep_alpha.tp.weight");
        const float* vw = S.get(B, ga + ".sep_value.lin.tp.weight");
        const float* w2 = S.get(B, ga + ".sep_value.dtp.tp.weight");
        // flat DTP weight index of sorted channel k of block l
        auto wflat = [&](int l, int k) {
            for (int p = 0; p < dtp_num_paths<L>(); ++p) {
                const PathInfo pi = dtp_path<L>(p);
                if (pi.l3 == l && k >= pi.kofs && k < pi.kofs + pi.mul1) return pi.wstart + (k - pi.kofs);
            }
            return -1;
        };
        size_t lofs[4] = {0, 0, 0, 0}, vofs[4] = {0, 0, 0, 0};
        int lrows[4] = {0, 0, 0, 0}, vrows[4] = {0, 0, 0, 0};
        {
            size_t lo = 0, vo = 0;
            for (int l = 0; l <= L; ++l) {
                lofs[l] = lo; vofs[l] = vo;
                lo += (size_t)dtp_k<L>(l) * (l == 0 ? lin0_rows<L>() : mul_of(l));
                vo += (size_t)dtp_k<L>(l) * mul_of(l);
                lrows[l] = l == 0 ? alpha_row0<L>() + mul_of(0) : mul_of(l);
                vrows[l] = mul_of(l);
            }
        }
        const int a0 = alpha_row0<L>(), O0 = lin0_rows<L>();
        // sep_act.lin rows, then (from row a0) the sep_alpha rows, share the l3 = 0 chunks
        // SO2: the reference coefficient of the chunk's path (dedf_tables.h::kSo2Ref) rides on the lin weights of the l3 >= 1 blocks; the value
        // stream folds its coefficients per class (pack_sval_stream), val_w_max bounds them for the power-of-two scale
        auto path_of = [&](int l, int k) { return dtp_path<L>(dtp_path_of_row<L>(wflat(l, k))); };
        auto lin_w = [&](int l, int oo, int k) {
            if (l == 0) return oo < O0 ? lw[lofs[0] + (size_t)k * O0 + oo] : (oo >= a0 ? aw[(size_t)k * mul_of(0) + (oo - a0)] : 0.0f);
            return lw[lofs[l] + (size_t)k * mul_of(l) + oo] * (SO2 ? so2_ref<L>(path_of(l, k)) : 1.0f);
        };
        auto val_w = [&](int l, int oo, int k) { return vw[vofs[l] + (size_t)k * mul_of(l) + oo] * w2[wflat(l, k)]; };
        auto so2_cmax = [&](int l, int k) {
            float m = 1.0f;
            if (SO2) { const PathInfo pi = path_of(l, k); m = 0.0f; for (int t = 0; t < kSo2NT[pi.l1][pi.l2][pi.l3]; ++t) m = std::fmax(m, std::fabs(kSo2C[pi.l1][pi.l2][pi.l3][t])); }
            return m;
        };
        int su = 8 - kActHeadroomBits;
        o.est_value = 1.0f;
        if (!un) {
            // gated features u' = Gate(lin(w x')): typical magnitude from the rms row norms of lin; the value from those of the value linear
            float eu = 0.0f;
            const float eb = est_we * est_msg * (SO2 ? 1.0f : 1.5f);      // (edge frame: the path coefficient rides on lin_w already)
            for (int l = 0; l <= L; ++l) eu = std::fmax(eu, 1.85f * eb * rms_row_norm(l == 0 ? O0 : mul_of(l), dtp_k<L>(l), [&](int oo, int k) { return lin_w(l, oo, k); }));
            su = act_exponent(eu);
            float ev = 0.0f;
            for (int l = 0; l <= L; ++l) ev = std::fmax(ev, eu * rms_row_norm(mul_of(l), dtp_k<L>(l), [&](int oo, int k) { return val_w(l, oo, k) * so2_cmax(l, k); }));
            o.est_value = ev;
        }
        o.s3 = s3; o.su = su;
        int sl[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0};
        for (int l = 0; l <= L; ++l) {
            float ml = 0.0f, mv = 0.0f;
            for (int k = 0; k < dtp_k<L>(l); ++k) {
                for (int oo = 0; oo < lrows[l]; ++oo) ml = std::fmax(ml, std::fabs(lin_w(l, oo, k)));
                for (int oo = 0; oo < vrows[l]; ++oo) mv = std::fmax(mv, std::fabs(val_w(l, oo, k)) * so2_cmax(l, k));
            }
            sl[l] = pow2_scale(ml, 0, 20); sv[l] = pow2_scale(mv, 0, 20);
            o.c_lin[l] = std::ldexp(1.0f, -(sl[l] + s3));
            o.c_val[l] = std::ldexp(1.0f, -(sv[l] + su));
        }
        o.u_scale = std::ldexp(1.0f, su);
        o.o_S_lin = im.push(pack_dtp_stream<L, SO2>(r0_tiles<L>(), lrows, [&](int l, int oo, int k) { return std::ldexp(lin_w(l, oo, k), sl[l]); }));
        auto val_at = [&](int p, int oo, int u) {
            const PathInfo pi = dtp_path<L>(p);
            return std::ldexp(val_w(pi.l3, oo, pi.kofs + u), sv[pi.l3]);
        };
        if constexpr (SO2) o.o_S_val = im.push(pack_sval_stream<L>(val_at)); else o.o_S_val = im.push(pack_val_stream<L>(val_at));
        {
            const float* lb = S.get(B, ga + ".sep_act.lin.bias.0");
            const float* ab = S.get(B, ga + ".sep_alpha.bias.0");
            const float f0 = std::ldexp(1.0f, sl[0] + s3);
            o.o_b_r0 = im.push(pack_rows(a0 + mul_of(0), [&](int i) { return (i < O0 ? lb[i] : (i >= a0 ? ab[i - a0] : 0.0f)) * f0; }));
        }
        {
            const float* vb = S.get(B, ga + ".sep_value.lin.bias.0");
            const float f0 = std::ldexp(1.0f, sv[0] + su);
            o.o_b_val0 = im.push(pack_rows(mul_of(0), [&](int i) { return vb[i] * f0; }));
        }
        o.o_alpha_dot = im.push(rows(mul_of(0), S.get(B, ga + ".alpha_dot")));
    }
}

// est_z: typical magnitude of the aggregate the kernel reads (EdgeOffsets::est_value); <= 0: unknown (UNet layers) -> the fixed 2^kNodeBShift
template <int L>
inline void pack_node(const dedf_config& c, const ParamSpec& S, const float* B, Image& im, NodeOffsets& o, float est_z = 0.0f) {
    // Every GEMM of the node kernel runs on split-fp16 MFMAs: hi / lo images (dedf_layout.h::pack_A_h), each matrix scaled by
    // its own power of two (pow2_scale); biases carry the matrix scale times the fixed B-operand scale 2^kNodeBShift.
    const bool un = c.unet_layer != 0;
    const std::string blk = un ? "gnn" : "key_tensor_field.gnn_block_init", ga = blk + ".ga", post = blk + (un ? ".norm_2" : ".post_norm");
    const float* pw = S.get(B, ga + ".proj.tp.weight");
    const float* lnw = S.get(B, post + ".affine_weight");
    const float* f1w = S.get(B, blk + ".ffn.fctp_1.tp.weight");
    const float* f2w = S.get(B, blk + ".ffn.fctp_2.tp.weight");
    // pack one matrix (O rows, K columns fed by a producer's accumulator tiles); returns its power-of-two exponent
    auto push_h = [&](int O, int K, auto W, int& off_h, int& off_l) {
        float mx = 0.0f;
        for (int oo = 0; oo < O; ++oo) for (int k = 0; k < K; ++k) mx = std::fmax(mx, std::fabs(W(oo, k)));
        const int sh = pow2_scale(mx, 0, 20);
        std::vector<float> ih, il;
        pack_A_h(O, cdiv(K, 16), [&](int oo, int k) { return std::ldexp(W(oo, k), sh); }, [&](int cc, int j, int h) { return chain_k(K, cc, j, h); }, ih, il);
        off_h = im.push(ih); off_l = im.push(il);
        return sh;
    };
    int s_proj0 = 0, s_f10 = 0, s_f20 = 0;
    size_t po = 0, lo = 0, f1o = 0, f2o = 0;
    // B-operand exponents per stage from the weights (act_exponent): typical magnitudes propagated from the aggregate through proj, the LayerNorm
    // (which resets the scale to its affine weights), the two FFN layers and the first stage of the score tensor products
    const bool adaptive = !un && est_z > 0.0f;
    int ez = kNodeBShift, en = kNodeBShift, eh[4] = {kNodeBShift, kNodeBShift, kNodeBShift, kNodeBShift}, ef = kNodeBShift, et[2] = {kNodeBShift, kNodeBShift};
    float est_fld = 0.0f;
    if (adaptive) {
        ez = act_exponent(est_z);
        const float* lnb = S.get(B, post + ".affine_bias");
        double n2 = 0.0; int nn = 0;
        for (int l = 0, q = 0; l <= L; ++l) for (int i = 0; i < mul_of(l); ++i, ++q) { n2 += (double)lnw[q] * lnw[q] + (l == 0 ? (double)lnb[i] * lnb[i] : 0.0); ++nn; }
        const float est_n = (float)std::sqrt(n2 / nn);
        en = act_exponent(est_n);
        size_t po_ = 0, f1_ = 0, f2_ = 0;
        for (int l = 0; l <= L; ++l) {
            const int m = mul_of(l), O1 = (l == 0 ? f1_rows0<L>() : hid_of(l)), Kh = hid_of(l);
            const float *Wp_ = pw + po_, *W1_ = f1w + f1_, *W2_ = f2w + f2_;
            const float est_emb = est_z * rms_row_norm(m, m, [&](int oo, int k) { return Wp_[k * m + oo]; });
            const float est_h = 1.85f * est_n * rms_row_norm(l == 0 ? hid_of(0) : O1, m, [&](int oo, int k) { return W1_[k * O1 + oo]; });
            eh[l] = act_exponent(est_h);
            est_fld = std::fmax(est_fld, est_emb + est_h * rms_row_norm(m, Kh, [&](int oo, int k) { return W2_[k * m + oo]; }));
            po_ += (size_t)m * m; f1_ += (size_t)m * O1; f2_ += (size_t)Kh * m;
        }
        ef = act_exponent(est_fld);
    }
    for (int l = 0; l <= L; ++l) {
        const int m = mul_of(l), O1 = (l == 0 ? f1_rows0<L>() : hid_of(l)), Kh = hid_of(l);
        const float* Wp = pw + po;
        const int sp = push_h(m, m, [&](int oo, int k) { return Wp[k * m + oo]; }, o.o_A_proj[l], o.o_A_proj_l[l]);
        o.o_ln_w[l] = im.push(pack_rows(m, [&](int i) { return lnw[lo + i]; }));
        const float* W1 = f1w + f1o;
        const int s1 = push_h(O1, m, [&](int oo, int k) { return W1[k * O1 + oo]; }, o.o_A_f1[l], o.o_A_f1_l[l]);
        const float* W2 = f2w + f2o;
        const int s2 = push_h(m, Kh, [&](int oo, int k) { return W2[k * m + oo]; }, o.o_A_f2[l], o.o_A_f2_l[l]);
        o.sc.proj[l] = std::ldexp(1.0f, -(sp + ez));
        o.sc.f1[l] = std::ldexp(1.0f, -(s1 + en));
        o.sc.f2[l] = std::ldexp(1.0f, -(s2 + eh[l]));
        if (l == 0) { s_proj0 = sp; s_f10 = s1; s_f20 = s2; }
        po += (size_t)m * m; lo += m; f1o += (size_t)m * O1; f2o += (size_t)Kh * m;
    }
    auto rows_s = [&](int O, const float* v, int sh) { return pack_rows(O, [&](int i) { return std::ldexp(v[i], sh); }); };
    o.o_b_proj0 = im.push(rows_s(mul_of(0), S.get(B, ga + ".proj.bias.0"), s_proj0 + ez));
    o.o_ln_b0 = im.push(rows_s(mul_of(0), S.get(B, post + ".affine_bias"), 0));
    o.o_b_f1 = im.push(rows_s(f1_rows0<L>(), S.get(B, blk + ".ffn.fctp_1.bias.0"), s_f10 + en));
    o.o_b_f2 = im.push(rows_s(mul_of(0), S.get(B, blk + ".ffn.fctp_2.bias.0"), s_f20 + eh[0]));
    int tp = 0;
    if (!c.ebm && !un) for (const char* nm : {"lin_vel_tp", "ang_vel_tp"}) {
        const std::string p = std::string(nm);
        const float* dw = S.get(B, p + ".dtp.tp.weight");
        if (adaptive) {      // contracted first-stage outputs (x the rotated query features, taken as O(1): they are the caller's data)
            float est_t = 0.0f;
            for (int q = 0; q < stp_num_paths<L>(); ++q) {
                const PathInfo pi = stp_path<L>(q);
                const float* W = dw + pi.wstart;
                est_t = std::fmax(est_t, 1.5f * est_fld * rms_row_norm(pi.mul1, pi.mul2, [&](int u, int v) { return W[u * pi.mul2 + v]; }));
            }
            et[tp] = act_exponent(est_t);
        }
        for (int q = 0; q < stp_num_paths<L>(); ++q) {
            const PathInfo pi = stp_path<L>(q);
            const float* W = dw + pi.wstart;
            const int sh = push_h(pi.mul1, pi.mul2, [&](int u, int v) { return W[u * pi.mul2 + v]; }, o.o_A_s[tp][q], o.o_A_s_l[tp][q]);
            o.sc.s[tp][q] = std::ldexp(1.0f, -(sh + ef));
        }
        // final LinearRS: K walks the 16-channel chunks of the TP output (stp_chunk_index), rows = the 32 gates (l3 = 0) / 32 1e channels
        const float* lw = S.get(B, p + ".lin.tp.weight");
        const int n1 = mul_of(1), O0 = 1 + n1;
        const float* Wl[2] = {lw, lw + (size_t)stp_k<L>(0) * O0};
        int shl[2] = {0, 0};
        for (int l3 = 0; l3 < 2; ++l3) {
            std::vector<int> base;                      // first sorted channel of every chunk
            for (int q = 0; q < stp_num_paths<L>(); ++q) {
                const PathInfo pi = stp_path<L>(q);
                if (pi.l3 == l3) for (int cu = 0; cu < pi.mul1 / 16; ++cu) base.push_back(pi.kofs + 16 * cu);
            }
            auto Wat = [&](int oo, int k) { return l3 == 0 ? Wl[0][k * O0 + 1 + oo] : Wl[1][k * n1 + oo]; };   // row 0 (the unused 1x0e) is dropped
            float mx = 0.0f;
            for (int oo = 0; oo < n1; ++oo) for (int k = 0; k < stp_k<L>(l3); ++k) mx = std::fmax(mx, std::fabs(Wat(oo, k)));
            shl[l3] = pow2_scale(mx, 0, 20);
            std::vector<float> ih, il;
            pack_A_h(n1, (int)base.size(), [&](int oo, int k) { return std::ldexp(Wat(oo, k), shl[l3]); },
                     [&](int cc, int j, int h) { return base[cc] + chunk_row(8 * h + j); }, ih, il);
            o.o_A_sl[tp][l3] = im.push(ih); o.o_A_sl_l[tp][l3] = im.push(il);
            o.sc.sl[tp][l3] = std::ldexp(1.0f, -(shl[l3] + et[tp]));
        }
        const float* lb = S.get(B, p + ".lin.bias.0");
        o.o_b_sl[tp] = im.push(pack_rows(n1, [&](int i) { return std::ldexp(lb[1 + i], shl[0] + et[tp]); }));
        ++tp;
    }
    o.sc.bz = std::ldexp(1.0f, ez); o.sc.bn = std::ldexp(1.0f, en); o.sc.bf = std::ldexp(1.0f, ef);
    for (int l = 0; l < 4; ++l) o.sc.bh[l] = std::ldexp(1.0f, eh[l]);
    o.sc.bt[0] = std::ldexp(1.0f, et[0]); o.sc.bt[1] = std::ldexp(1.0f, et[1]);
    // the kernel addresses the image through the compile-time layout: it must be what was just built
    constexpr NodeLayout<L> nl = kNodeLayout<L>;
    bool same = o.o_b_proj0 == nl.b_proj0 && o.o_ln_b0 == nl.ln_b0 && o.o_b_f1 == nl.b_f1 && o.o_b_f2 == nl.b_f2;
    for (int l = 0; l <= L; ++l)
        same = same && o.o_A_proj[l] == nl.A_proj[l] && o.o_A_proj_l[l] == nl.A_proj_l[l] && o.o_ln_w[l] == nl.ln_w[l] && o.o_A_f1[l] == nl.A_f1[l] &&
               o.o_A_f1_l[l] == nl.A_f1_l[l] && o.o_A_f2[l] == nl.A_f2[l] && o.o_A_f2_l[l] == nl.A_f2_l[l];
    for (int t = 0; t < tp; ++t) {
        for (int q = 0; q < stp_num_paths<L>(); ++q) same = same && o.o_A_s[t][q] == nl.A_s[t][q] && o.o_A_s_l[t][q] == nl.A_s_l[t][q];
        same = same && o.o_A_sl[t][0] == nl.A_sl[t][0] && o.o_A_sl_l[t][0] == nl.A_sl_l[t][0] && o.o_A_sl[t][1] == nl.A_sl[t][1] &&
               o.o_A_sl_l[t][1] == nl.A_sl_l[t][1] && o.o_b_sl[t] == nl.b_sl[t];
    }
    o.layout_ok = same;
}

}  // namespace dedf
