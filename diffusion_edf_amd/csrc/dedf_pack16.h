// Host side of the 16-edge tile (dedf_edge16.h): the A-operand images of v_mfma_f32_16x16x32_f16 in the order the kernel consumes them.
// Same arithmetic as dedf_pack.h::pack_edge (3-term split-fp16 products, the same power-of-two operand scales); only the lane layout differs.
#pragma once
#include "dedf_net16.h"
#include "dedf_pack.h"

namespace dedf {

struct Edge16Offsets {
    int o_A3, o_A3_l, o_off3;         // last radial layer: images [tile][chunk][lane][8 halves], offsets in natural walk-row order
    int o_S_lin, o_b0;                // sep_act.lin + sep_alpha stream (slot = hi | lo image of one (chunk, output tile)), bias rows (natural order)
    int o_S_val, o_bval0, o_adot;     // sep_value.lin (shared DTP weights folded in), its bias, alpha_dot
    float w_unscale, u_scale, c_lin[4], c_val[4];
    bool ok;
};

// one operand image: 64 lanes x 8 halves; W(lane & 15, lane >> 4, j) -> value
template <class WAt>
inline void put_op16(std::vector<uint16_t>& hi, std::vector<uint16_t>& lo, size_t op, WAt W) {
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const float w = W(lane & 15, lane >> 4, j);
            const _Float16 hh = (_Float16)w;
            const _Float16 ll = (_Float16)(w - (float)hh);
            const size_t idx = (op * 64 + lane) * 8 + j;
            __builtin_memcpy(&hi[idx], &hh, 2);
            __builtin_memcpy(&lo[idx], &ll, 2);
        }
}
inline std::vector<float> halves_to_floats(const std::vector<uint16_t>& h) {
    std::vector<float> out(h.size() / 2);
    __builtin_memcpy(out.data(), h.data(), h.size() * 2);
    return out;
}

template <int L>
inline void pack_edge16(const dedf_config& c, const ParamSpec& S, const float* B, Image& im, Edge16Offsets& o, int msg_exp = 0) {
    // msg_exp: log2 of the power of two the source message is stored with (EdgeOffsets::msg_scale): the lin accumulators carry it
    o.ok = false;
    if (c.unet_layer || c.half_gemm || c.fc_neurons[2] != 64) return;          // instantiated for the score heads with the [., 128, 64] radial MLP
    const std::string ga = "key_tensor_field.gnn_block_init.ga", rad = ga + ".sep_act.dtp_rad.";
    constexpr int WN = dtp_wn<L>(), NT = w16_tiles<L>(), H2 = 64;
    static_assert(NT * 16 == WN, "16-channel weight tiles");
    // ---- last radial layer: rows in walk order; K = 64 = two chunks whose element (g, j) is activation 32 c + 8 g + j (the table rows are
    //      read in that order, dedf_edge16.h)
    int s3 = 0;
    {
        const float* W3 = S.get(B, rad + "net.6.weight");
        const float* off3 = S.get(B, rad + "offset");
        float m3 = 0.0f;
        for (int i = 0; i < WN * H2; ++i) m3 = std::fmax(m3, std::fabs(W3[i]));
        for (int i = 0; i < WN; ++i) m3 = std::fmax(m3, std::fabs(off3[i]));
        s3 = pow2_scale(m3, -16, 8 + kActHeadroomBits) - kActHeadroomBits;          // as dedf_pack.h::pack_edge
        const float f3 = std::ldexp(1.0f, s3);
        std::vector<uint16_t> hi((size_t)NT * 2 * 512), lo(hi.size());
        for (int t = 0; t < NT; ++t)
            for (int cc = 0; cc < 2; ++cc)
                put_op16(hi, lo, (size_t)t * 2 + cc, [&](int r, int g, int j) { return W3[(size_t)w16_weight_row<L>(16 * t + r) * H2 + 32 * cc + 8 * g + j] * f3; });
        o.o_A3 = im.push(halves_to_floats(hi)); o.o_A3_l = im.push(halves_to_floats(lo));
        std::vector<float> offs(WN);
        for (int r = 0; r < WN; ++r) offs[r] = off3[w16_weight_row<L>(r)] * f3;
        o.o_off3 = im.push(offs);
        o.w_unscale = std::ldexp(1.0f, -s3);
    }
    // ---- sep_act.lin (+ sep_alpha on the l3 = 0 chunks) and sep_value.lin with the shared depth-wise weights folded in
    const float* lw = S.get(B, ga + ".sep_act.lin.tp.weight");
    const float* aw = S.get(B, ga + ".sep_alpha.tp.weight");
    const float* vw = S.get(B, ga + ".sep_value.lin.tp.weight");
    const float* w2 = S.get(B, ga + ".sep_value.dtp.tp.weight");
    auto wflat = [&](int l, int k) {
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            const PathInfo pi = dtp_path<L>(p);
            if (pi.l3 == l && k >= pi.kofs && k < pi.kofs + pi.mul1) return pi.wstart + (k - pi.kofs);
        }
        return -1;
    };
    size_t lofs[4] = {0, 0, 0, 0}, vofs[4] = {0, 0, 0, 0};
    {
        size_t lo_ = 0, vo = 0;
        for (int l = 0; l <= L; ++l) {
            lofs[l] = lo_; vofs[l] = vo;
            lo_ += (size_t)dtp_k<L>(l) * (l == 0 ? lin0_rows<L>() : mul_of(l));
            vo += (size_t)dtp_k<L>(l) * mul_of(l);
        }
    }
    const int O0 = lin0_rows<L>();          // rows [0, O0): lin scalars + gates, [O0, O0 + 64): alpha -- no padding in this row space
    auto lin_w = [&](int l, int oo, int k) {
        if (l == 0) return oo < O0 ? lw[lofs[0] + (size_t)k * O0 + oo] : aw[(size_t)k * mul_of(0) + (oo - O0)];
        return lw[lofs[l] + (size_t)k * mul_of(l) + oo];
    };
    auto val_w = [&](int l, int oo, int k) { return vw[vofs[l] + (size_t)k * mul_of(l) + oo] * w2[wflat(l, k)]; };
    const int su = 8 - kActHeadroomBits;
    int sl[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0};
    for (int l = 0; l <= L; ++l) {
        float ml = 0.0f, mv = 0.0f;
        const int lr = lin_tiles16<L>(l) * 16;
        for (int k = 0; k < dtp_k<L>(l); ++k) {
            for (int oo = 0; oo < lr; ++oo) ml = std::fmax(ml, std::fabs(lin_w(l, oo, k)));
            for (int oo = 0; oo < mul_of(l); ++oo) mv = std::fmax(mv, std::fabs(val_w(l, oo, k)));
        }
        sl[l] = pow2_scale(ml, 0, 20); sv[l] = pow2_scale(mv, 0, 20);
        o.c_lin[l] = std::ldexp(1.0f, -(sl[l] + s3 + msg_exp));
        o.c_val[l] = std::ldexp(1.0f, -(sv[l] + su));
    }
    o.u_scale = std::ldexp(1.0f, su);
    {   // lin stream: group l3, chunk q (two walk tiles), output tile To
        std::vector<uint16_t> hi((size_t)lin16_slots<L>() * 512), lo(hi.size());
        std::vector<float> img((size_t)lin16_slots<L>() * 512);
        for (int l3 = 0; l3 <= L; ++l3)
            for (int q = 0; q < w16_group_chunks<L>(l3); ++q)
                for (int To = 0; To < lin_tiles16<L>(l3); ++To) {
                    std::vector<uint16_t> h1(512), l1(512);
                    put_op16(h1, l1, 0, [&](int r, int g, int j) {
                        const int tg = 2 * q + chain16_tile(j);                 // tile inside the group
                        if (tg >= w16_group_tiles<L>(l3)) return 0.0f;          // the odd last tile of a group: half a chunk
                        const int t = kWalk16<L>.grp0[l3] + tg;
                        return std::ldexp(lin_w(l3, 16 * To + r, w16_channel<L>(t, chain16_row(g, j))), sl[l3]);
                    });
                    const size_t slot = (size_t)lin16_slot<L>(l3, q) + To;
                    __builtin_memcpy(&img[slot * 512], h1.data(), 1024);
                    __builtin_memcpy(&img[slot * 512 + 256], l1.data(), 1024);
                }
        o.o_S_lin = im.push(img);
        const float* lb = S.get(B, ga + ".sep_act.lin.bias.0");
        const float* ab = S.get(B, ga + ".sep_alpha.bias.0");
        const float f0 = std::ldexp(1.0f, sl[0] + s3 + msg_exp);
        std::vector<float> b0(lin0_tiles16<L>() * 16);
        for (int i = 0; i < (int)b0.size(); ++i) b0[i] = (i < O0 ? lb[i] : ab[i - O0]) * f0;
        o.o_b0 = im.push(b0);
    }
    {   // value stream: path p (creation order inside its l3 group), input chunk c, output tile To
        std::vector<float> img((size_t)val16_slots<L>() * 512);
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            const PathInfo pi = dtp_path<L>(p);
            for (int cc = 0; cc < feat_chunks16(pi.l1); ++cc)
                for (int To = 0; To < val_tiles16(pi.l3); ++To) {
                    std::vector<uint16_t> h1(512), l1(512);
                    put_op16(h1, l1, 0, [&](int r, int g, int j) {
                        const int u = feat16_channel(pi.l1, cc, g, j);
                        return u < 0 ? 0.0f : std::ldexp(val_w(pi.l3, 16 * To + r, pi.kofs + u), sv[pi.l3]);
                    });
                    const size_t slot = (size_t)val16_slot<L>(p, cc, To);
                    __builtin_memcpy(&img[slot * 512], h1.data(), 1024);
                    __builtin_memcpy(&img[slot * 512 + 256], l1.data(), 1024);
                }
        }
        o.o_S_val = im.push(img);
        const float* vb = S.get(B, ga + ".sep_value.lin.bias.0");
        const float f0 = std::ldexp(1.0f, sv[0] + su);
        std::vector<float> bv(mul_of(0));
        for (int i = 0; i < mul_of(0); ++i) bv[i] = vb[i] * f0;
        o.o_bval0 = im.push(bv);
        const float* ad = S.get(B, ga + ".alpha_dot");
        o.o_adot = im.push(std::vector<float>(ad, ad + mul_of(0)));
    }
    o.ok = true;
}

}  // namespace dedf
