// Fused per-edge pipeline on a 16-EDGE tile (one wave = 16 edges of one scale), v_mfma_f32_16x16x32_f16, built for TWO waves per SIMD.
// Same arithmetic as dedf_edge.h (3-term split-fp16 GEMMs with the same operand scales, lane-local Clebsch-Gordan contractions, segment
// softmax partials, the sampler's radial table) -- see dedf_net16.h for the layout and why this tiling exists.  Sampler only (MODE 1 of
// dedf_edge.h: the radial network's front comes from the table); lmax 2, score head with the [128, 128, 64] radial MLP.  A tile whose lengths
// leave the table (or whose scale failed the table's accuracy check) is NOT evaluated here: the wave raises `redo` and the host re-runs the
// step's edge stage on the 32-edge kernel.
#pragma once
#include "dedf_edge.h"
#include "dedf_net16.h"

namespace dedf {

struct Edge16Params {
    EdgeParams E;                     // graph, source message, cut-offs, radial table, output records (its weight offsets are the 32-edge image's: unused)
    const float* W; uint32_t W_bytes; // the 16-edge weight image (dedf_pack16.h)
    int o_A3, o_A3_l, o_off3, o_S_lin, o_b0, o_S_val, o_bval0, o_adot;
    float w_unscale, u_scale, c_lin[4], c_val[4];
    int* redo;                        // set to 1 when some tile could not take its front from the table
};

DEDF_DEV f32x4 mfma16(h8 a, h8 b, f32x4 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    (void)a; (void)b; return c;
#endif
}
// 3-term split product  acc += (ah + al) (bh + bl)  without the lo x lo term
DEDF_DEV f32x4 mfma16x3(const f32x4& ah, const f32x4& al, const HL& b, f32x4 acc) {
    acc = mfma16(__builtin_bit_cast(h8, ah), b.hi, acc);
    acc = mfma16(__builtin_bit_cast(h8, ah), b.lo, acc);
    acc = mfma16(__builtin_bit_cast(h8, al), b.hi, acc);
    return acc;
}

template <int L> struct Rows16 {      // per-row vectors in this wave's LDS, natural row order
    static constexpr int off3 = 0, b0 = off3 + dtp_wn<L>(), bval0 = b0 + lin0_tiles16<L>() * 16, adot = bval0 + 64, total = adot + 64;
};
// parked gated features (B operands of the value GEMMs): full 16-byte slots for degrees 0, 1 (hi, lo per chunk), 8-byte slots for degree 2
// (a 16-channel block fills half a K = 32 chunk: 4 valid halves per lane)
template <int L> struct Park16 {
    static constexpr int n_full = park16_slot<L>(2, 0, 0);                    // chunks of degrees 0 and 1: 5
    static constexpr int full_bytes = n_full * 2 * 1024;
    static constexpr int n_half = L >= 2 ? 5 : 0;
    static constexpr int bytes = full_bytes + n_half * 2 * 512;
};

template <int L>
DEDF_DEV void edge16_tile(const Edge16Params& Q, const Buf& wb, int lane, int scale, int e0, int n_valid, float* __restrict__ rows, char* __restrict__ park) {
    static_assert(L == 2, "the 16-edge tile is instantiated for lmax 2");
    const EdgeParams& P = Q.E;
    constexpr int D = feat_dim<L>(), REC = edge_rec<L>(), NT = w16_tiles<L>();
    using RW = Rows16<L>;
    const int e = lane & 15, g = lane >> 4, lane16 = lane * 16;
    // every weight operand (one 1 KiB image per wave) comes through here.  -DDEDF_E16_LDS_TIMING: the images are read from this wave's LDS
    // instead (WRONG results: finite garbage) -- the timing experiment behind DESIGN.md's "why the 16-edge tile loses": what the kernel would
    // cost if the weight stream did not go through the CU's vector L1.
    auto wld4 = [&](int off) {
#if defined(DEDF_E16_LDS_TIMING)
        int a = lane16 + Park16<L>::bytes + (off & 0);          // a zeroed KiB behind the parked features; opaque: no CSE of the reads
        asm volatile("" : "+v"(a));
        return *reinterpret_cast<const f32x4*>(park + a);
#else
        return bld4(wb, lane16, off);
#endif
    };
    const bool valid = e < n_valid;
    const int ei = e0 + (valid ? e : 0);
    const int src = P.edge_src[ei], dst = P.edge_dst[ei];

    // ---- geometry (graph_parser.py:159-215), as dedf_edge.h ------------------------------------------------------------------------
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
        const float rho = ux * ux + uz * uz;
        Y.y2[0] = s5 * s3 * ux * uz * cns;
        Y.y2[1] = s5 * s3 * ux * uy * cns;
        Y.y2[2] = s5 * (uy * uy - 0.5f * rho) * cns;
        Y.y2[3] = s5 * s3 * uy * uz * cns;
        Y.y2[4] = s5 * s3 * 0.5f * (uz * uz - ux * ux) * cns;
    }

    // ---- front of the radial network from the table: activation 32 c + 8 g + j of layer 2 (4-point Lagrange, dedf_edge.h MODE 1) ---------
    const float pos = len * P.rtab_inv_step[scale];
    const bool accurate = __builtin_bit_cast(float, P.rtab_err[scale]) <= P.rtab_err_bound[scale];
    const bool tab = accurate && __all(!valid || pos < (float)P.rtab_n[scale]) != 0;
    if (!tab) { if (lane == 0) *Q.redo = 1; return; }
    HL r2s[2];
    {
        const float ps = valid ? pos : 0.0f;
        const int i0 = (int)ps;
        const float u = ps - (float)i0;
        const float um = u - 1.0f, up = u + 1.0f, u2 = u - 2.0f;
        const float tw[4] = {-(1.0f / 6) * u * um * u2, 0.5f * up * um * u2, -0.5f * up * u * u2, (1.0f / 6) * up * u * um};
        const Buf rtb = make_buf(P.rtab, P.rtab_bytes);
        // table row = [half 2][32]: position (half, 16 T + R) holds activation 32 T + rowmap(R, half); activation 32 c + 8 g + j sits at
        // half = j >> 2, position 16 c + 4 g + (j & 3): two 16-byte reads per (row, chunk)
        const int rv = (P.rtab_row0[scale] + i0) * 256 + g * 16;
        f32x4 t[4][2][2];
        static_for<4>([&]<int K>() { static_for<2>([&]<int c>() { static_for<2>([&]<int hf>() {
            t[K][c][hf] = bld4(rtb, rv, K * 256 + hf * 128 + c * 64);
        }); }); });
        static_for<2>([&]<int c>() {
            float v[8];
            static_for<2>([&]<int hf>() { static_for<4>([&]<int J>() {
                v[4 * hf + J] = (tw[0] * t[0][c][hf][J] + tw[1] * t[1][c][hf][J]) + (tw[2] * t[2][c][hf][J] + tw[3] * t[3][c][hf][J]);
            }); });
            r2s[c] = split8(v);
        });
    }

    // ---- layer 3 (-> per-edge TP weights, one 16-row tile at a time) fused with DTP #1 and the lin / sep_alpha GEMMs ----------------------
    f32x4 acc0[lin0_tiles16<L>()], acc1[3][2], acc2[5][1];
    static_for<lin0_tiles16<L>()>([&]<int T>() { acc0[T] = *reinterpret_cast<const f32x4*>(rows + RW::b0 + 16 * T + 4 * g); });
    const Buf msgb = make_buf(P.msg, P.msg_bytes);
    const int mvb = src * (D * 4);
    float logit[kHeads], g1[2][4], g2[4];
    const float cl0 = Q.c_lin[0], cl1 = Q.c_lin[1], cl2 = Q.c_lin[2], us = Q.u_scale;
    f32x4* const pk = reinterpret_cast<f32x4*>(park) + lane;                                  // full slots: [slot][lane] 16 B
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2* const pk2 = reinterpret_cast<f32x2*>(park + Park16<L>::full_bytes) + lane;         // half slots: [slot][lane] 8 B
    auto park_full = [&]<int Qs>(const float (&v)[8]) {
        const HL sp = split8(v);
        pk[(2 * Qs) * 64] = __builtin_bit_cast(f32x4, sp.hi);
        pk[(2 * Qs + 1) * 64] = __builtin_bit_cast(f32x4, sp.lo);
    };
    auto park_half = [&]<int Qs>(const float (&v4)[4]) {
        const float v[8] = {v4[0], v4[1], v4[2], v4[3], 0.0f, 0.0f, 0.0f, 0.0f};
        const HL sp = split8(v);
        const f32x4 h = __builtin_bit_cast(f32x4, sp.hi), l = __builtin_bit_cast(f32x4, sp.lo);
        pk2[(2 * Qs) * 64] = f32x2{h[0], h[1]};
        pk2[(2 * Qs + 1) * 64] = f32x2{l[0], l[1]};
    };
    auto finish_group = [&]<int l3>() {
        if constexpr (l3 == 0) {
            // attention logits: alpha rows 112 .. 175 = tiles 7 .. 10, one 16-channel head per tile
            constexpr int AT = lin0_rows<L>() / 16;
            static_for<kHeads>([&]<int hd>() {
                const f32x4 ad = *reinterpret_cast<const f32x4*>(rows + RW::adot + 16 * hd + 4 * g);
                float x4[4], s4[4];
                static_for<4>([&]<int R>() { x4[R] = acc0[AT + hd][R] * cl0; });
                sigmoid_stage<4>(x4, s4);
                float sum = 0.0f;
                static_for<4>([&]<int R>() { sum += (0.6f * x4[R] + 0.4f * x4[R] * (2.0f * s4[R] - 1.0f)) * kNormSlrelu * ad[R]; });
#if defined(__HIP_DEVICE_COMPILE__)
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
#endif
                logit[hd] = sum + logit0;
            });
            // Gate: SiLU on the 64 scalars (tiles 0-3 -> two parked chunks), sigmoid gates for the l = 1 (tiles 4, 5) and l = 2 (tile 6) channels
            static_for<2>([&]<int q>() {
                float v[8];
                static_for<8>([&]<int J>() { v[J] = acc0[2 * q + (J >> 2)][J & 3] * cl0; });
                silu_stage<8>(v);
                static_for<8>([&]<int J>() { v[J] = v[J] * (kNormSilu * us); });
                park_full.template operator()<park16_slot<L>(0, 0, q)>(v);
            });
            static_for<2>([&]<int t>() {
                float x4[4];
                static_for<4>([&]<int R>() { x4[R] = acc0[4 + t][R] * cl0; });
                sigmoid_stage<4>(x4, g1[t]);
                static_for<4>([&]<int R>() { g1[t][R] = g1[t][R] * (kNormSigmoid * (cl1 * us)); });
            });
            {
                float x4[4];
                static_for<4>([&]<int R>() { x4[R] = acc0[6][R] * cl0; });
                sigmoid_stage<4>(x4, g2);
                static_for<4>([&]<int R>() { g2[R] = g2[R] * (kNormSigmoid * (cl2 * us)); });
            }
        } else if constexpr (l3 == 1) {
            static_for<3>([&]<int K>() {
                float v[8];
                static_for<8>([&]<int J>() { v[J] = acc1[K][J >> 2][J & 3] * g1[J >> 2][J & 3]; });
                park_full.template operator()<park16_slot<L>(1, K, 0)>(v);
            });
        } else {
            static_for<5>([&]<int K>() {
                float v[4];
                static_for<4>([&]<int J>() { v[J] = acc2[K][0][J] * g2[J]; });
                park_half.template operator()<K>(v);
            });
        }
    };
    static_for<3>([&]<int K>() { static_for<2>([&]<int t>() { acc1[K][t] = f32x4{0, 0, 0, 0}; }); });
    static_for<5>([&]<int K>() { acc2[K][0] = f32x4{0, 0, 0, 0}; });

    // weight-image offsets, re-materialised per tile (dedf_dev.h::opaque_s: LICM otherwise hoists hundreds of `offset + constant` scalars out of
    // the persistent tile loop and spills them to VGPR lanes)
    const int oA3 = opaque_s(Q.o_A3), oA3l = opaque_s(Q.o_A3_l), oSl = opaque_s(Q.o_S_lin);
    // Software pipeline over the weight tiles.  Region t: requests the layer-3 operands of tile t + 2 and the source-message rows of tile
    // t + 1; runs the layer-3 MFMAs of tile t + 1 (two independent accumulators, one per K-chunk, terms interleaved: a 16x16x32 MFMA on the
    // accumulator of its predecessor waits out the predecessor's whole latency); does the Clebsch-Gordan work of tile t with the weights the
    // previous region produced; and, when t completes a K = 32 chunk, that chunk's lin / sep_alpha MFMAs (term-major over the output tiles).
    struct A3Ops { f32x4 ah[2], al[2]; };
    struct XOps16 { f32x4 x[2 * L + 1]; };
    auto load_a3 = [&]<int t>() {
        A3Ops o{};
        if constexpr (t < NT) static_for<2>([&]<int c>() {
            o.ah[c] = wld4((oA3 + (2 * t + c) * 256) * 4);
            o.al[c] = wld4((oA3l + (2 * t + c) * 256) * 4);
        });
        return o;
    };
    auto load_x = [&]<int t>() {
        XOps16 o{};
        if constexpr (t < NT) {
            constexpr PathInfo pi = kWalk16<L>.path[t];
            constexpr int l1 = pi.l1, d1 = 2 * l1 + 1;
            static_for<d1>([&]<int Qx>() { o.x[Qx] = bld4(msgb, mvb, (blk_off(l1) + (kWalk16<L>.u0[t] + 4 * g) * d1 + 4 * Qx) * 4); });
        }
        return o;
    };
    struct W2 { f32x4 a, b; };          // the two partial accumulators of a weight tile (K-chunk 0 + offset rows | K-chunk 1)
    auto run_l3 = [&]<int t>(const A3Ops& o) {
        W2 w{};
        if constexpr (t < NT) {
            w.a = *reinterpret_cast<const f32x4*>(rows + RW::off3 + 16 * t + 4 * g);
            w.b = f32x4{0, 0, 0, 0};
            w.a = mfma16(__builtin_bit_cast(h8, o.ah[0]), r2s[0].hi, w.a); w.b = mfma16(__builtin_bit_cast(h8, o.ah[1]), r2s[1].hi, w.b);
            w.a = mfma16(__builtin_bit_cast(h8, o.ah[0]), r2s[0].lo, w.a); w.b = mfma16(__builtin_bit_cast(h8, o.ah[1]), r2s[1].lo, w.b);
            w.a = mfma16(__builtin_bit_cast(h8, o.al[0]), r2s[0].hi, w.a); w.b = mfma16(__builtin_bit_cast(h8, o.al[1]), r2s[1].hi, w.b);
        }
        return w;
    };
    float vch[5][8];                  // the chunk being formed: [component][element j]
    A3Ops a3n = load_a3.template operator()<1>();
    XOps16 xn = load_x.template operator()<0>();
    W2 wn;
    {
        const A3Ops a30 = load_a3.template operator()<0>();
        wn = run_l3.template operator()<0>(a30);
    }
    static_for<NT>([&]<int t>() {
        constexpr PathInfo pi = kWalk16<L>.path[t];
        constexpr int l1 = pi.l1, l2 = pi.l2, l3 = pi.l3, d1 = 2 * l1 + 1, d3 = 2 * l3 + 1;
        constexpr int tg = t - kWalk16<L>.grp0[l3], q = tg / 2;
        constexpr bool last_of_group = t + 1 == kWalk16<L>.grp0[l3 + 1];
        constexpr bool chunk_done = (tg & 1) == 1 || last_of_group;
        constexpr int NTo = lin_tiles16<L>(l3), s0 = lin16_slot<L>(l3, q);
        const A3Ops a3c = a3n;          // operands of tile t + 1
        const XOps16 xc = xn;           // rows of tile t
        const W2 wc = wn;               // weights of tile t
        a3n = load_a3.template operator()<t + 2>();
        xn = load_x.template operator()<t + 1>();
        // A operands of this chunk's lin / sep_alpha GEMMs: the first ones requested now, used after the Clebsch-Gordan work below
        constexpr int NPF = chunk_done ? imin(NTo, 4) : 0;
        f32x4 lah[NPF > 0 ? NPF : 1], lal[NPF > 0 ? NPF : 1];
        static_for<NPF>([&]<int To>() {
            lah[To] = wld4((oSl + (s0 + To) * 512) * 4); lal[To] = wld4((oSl + (s0 + To) * 512 + 256) * 4);
        });
        wn = run_l3.template operator()<t + 1>(a3c);
        // Clebsch-Gordan products of this lane's 4 channels with the edge's harmonics, times the per-edge weights
        float xr[4 * d1];
        static_for<d1>([&]<int Qx>() { static_for<4>([&]<int i>() { xr[4 * Qx + i] = xc.x[Qx][i]; }); });
        using Cg = CG<l1, l2, l3>;
        float m[Cg::NM];
        Cg::make(Y.template get<l2>(), m);
        static_for<4>([&]<int r>() {
            float tt[d3];
            Cg::apply(&xr[r * d1], m, tt);
            const float wr = wc.a[r] + wc.b[r];
            static_for<d3>([&]<int K>() { vch[K][4 * (tg & 1) + r] = tt[K] * wr; });
        });
        if constexpr (chunk_done) {
            if constexpr ((tg & 1) == 0) static_for<d3>([&]<int K>() { static_for<4>([&]<int r>() { vch[K][4 + r] = 0.0f; }); });      // odd last tile: half a chunk
            HL b[d3];
            static_for<d3>([&]<int K>() { b[K] = split8(vch[K]); });
            // batches of up to 4 output tiles, terms outermost: consecutive MFMAs hit different accumulators
            static_for<cdiv(NTo, 4)>([&]<int B4>() {
                constexpr int T0 = 4 * B4, NB = imin(4, NTo - T0);
                f32x4 ah[NB], al[NB];
                static_for<NB>([&]<int n>() {
                    if constexpr (T0 + n < NPF) { ah[n] = lah[T0 + n]; al[n] = lal[T0 + n]; }
                    else { ah[n] = wld4((oSl + (s0 + T0 + n) * 512) * 4); al[n] = wld4((oSl + (s0 + T0 + n) * 512 + 256) * 4); }
                });
                static_for<3>([&]<int term>() { static_for<NB>([&]<int n>() { static_for<d3>([&]<int K>() {
                    const h8 av = __builtin_bit_cast(h8, term == 2 ? al[n] : ah[n]);
                    const h8 bv = term == 1 ? b[K].lo : b[K].hi;
                    if constexpr (l3 == 0) acc0[T0 + n] = mfma16(av, bv, acc0[T0 + n]);
                    else if constexpr (l3 == 1) acc1[K][T0 + n] = mfma16(av, bv, acc1[K][T0 + n]);
                    else acc2[K][T0 + n] = mfma16(av, bv, acc2[K][T0 + n]);
                }); }); });
            });
        }
        if constexpr (last_of_group) finish_group.template operator()<l3>();
        sched_fence();
    });

    // ---- joint-softmax partials over the runs of same-destination edges of this 16-lane row (dedf_edge.h, "joint-softmax partials") -------
    const int col = e;
    auto shf = [&](int addr, float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x))); };
    auto shi = [&](int addr, int x) { return __builtin_amdgcn_ds_bpermute(addr, x); };
    auto dpi = [&]<int CTRL, int ROWS, bool ZERO>(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWS, 0xf, ZERO); };
    auto dpf = [&]<int CTRL, int ROWS, bool ZERO>(float old, float x) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, ZERO));
    };
    int seg_start;
    bool seg_last;
    auto mkr = [&](int i) { return col - (1 << i) >= seg_start; };
    float pw[kHeads], inv_s[kHeads];
    {
        const int prev_dst = shi((lane - 1) * 4, dst);
        const bool head = col == 0 || dst != prev_dst || col == n_valid;
        seg_start = head ? col : 0;
        static_for<4>([&]<int i>() { seg_start = max(seg_start, dpi.template operator()<0x110 + (1 << i), 0xf, true>(0, seg_start)); });
        const int next_head = shi((lane + 1) * 4, head ? 1 : 0);
        const bool last_any = col == 15 || next_head != 0;
        seg_last = last_any && valid;
        int seg_end = last_any ? col : 15;
        static_for<4>([&]<int i>() { seg_end = min(seg_end, dpi.template operator()<0x100 + (1 << i), 0xf, false>(15, seg_end)); });
        const int end_addr = ((lane & 48) + seg_end) * 4;
        float lse[kHeads];
        static_for<kHeads>([&]<int h>() {
            float mx = logit[h];
            static_for<4>([&]<int i>() { const float tt = dpf.template operator()<0x110 + (1 << i), 0xf, false>(mx, mx); mx = mkr(i) ? fmaxf(mx, tt) : mx; });
            mx = shf(end_addr, mx);
            pw[h] = valid ? fexp(logit[h] - mx) : 0.0f;
            float sum = pw[h];
            static_for<4>([&]<int i>() { const float tt = dpf.template operator()<0x110 + (1 << i), 0xf, true>(0.0f, sum); sum += mkr(i) ? tt : 0.0f; });
            inv_s[h] = 1.0f / sum;
            lse[h] = mx + logf(sum);
        });
        if (seg_last && g == 0) st4(P.out + (size_t)(e0 + seg_start) * REC + D, f32x4{lse[0], lse[1], lse[2], lse[3]});
        const float wsrc = P.key_w != nullptr ? P.key_w[src] : 1.0f;
        static_for<kHeads>([&]<int h>() { pw[h] *= wsrc; });
    }
    int n_steps = 0;
    static_for<4>([&]<int i>() { if (__builtin_amdgcn_ballot_w64(mkr(i)) != 0) n_steps = i + 1; });
    float* const orec = P.out + (size_t)(e0 + seg_start) * REC;
    float* const drec = P.dbg_out != nullptr ? P.dbg_out + (size_t)ei * REC : nullptr;
    // x[n] = this lane's four channel values of record slot n (true units, already weighted): segmented inclusive scan along the row, the
    // segment's last lane stores the means
    auto scan_step = []<int i>(float (&x)[4], float m) {       // (operands as parameters: inline asm cannot name captured variables)
        if constexpr (i == 0) DEDF_FMAC_DPP4(x, m, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else if constexpr (i == 1) DEDF_FMAC_DPP4(x, m, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else if constexpr (i == 2) DEDF_FMAC_DPP4(x, m, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        else DEDF_FMAC_DPP4(x, m, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    };
    auto emit = [&]<int NS>(float (&x)[NS][4], const int (&rec_off)[NS], const float (&inv)[NS]) {
        static_for<4>([&]<int i>() {
            if (i < n_steps) {
                const float mk = mkr(i) ? 1.0f : 0.0f;
                static_for<NS>([&]<int n>() { scan_step.template operator()<i>(x[n], mk); });
            }
        });
        if (seg_last) static_for<NS>([&]<int n>() { st4(orec + rec_off[n], f32x4{x[n][0], x[n][1], x[n][2], x[n][3]} * inv[n]); });
    };

    // ---- sep_value: second depth-wise TP in output-side form (dedf_net.h::make_val_walk) + LinearRS -> value ---------------------------------
    // G_i[o] = sum_u W2[p,u,o] u[u,i] on the parked features (B operands as they are), then value[o,k] += (sum_j C_ijk Y_j) G_i[o] lane-locally
    static_for<L + 1>([&]<int l>() { static_for<2 * l + 1>([&]<int i>() {
        if constexpr (l == 0) opaque_v(Y.y0[i]); else if constexpr (l == 1) opaque_v(Y.y1[i]); else opaque_v(Y.y2[i]);
    }); });
    const float cv[3] = {Q.c_val[0], Q.c_val[1], Q.c_val[2]};
    auto load_B = [&]<int l, int i, int c>() {
        HL b;
        if constexpr (l < 2) {
            constexpr int s = park16_slot<L>(l, i, c);
            b.hi = __builtin_bit_cast(h8, pk[(2 * s) * 64]);
            b.lo = __builtin_bit_cast(h8, pk[(2 * s + 1) * 64]);
        } else {
            const f32x2 h = pk2[(2 * i) * 64], lo2 = pk2[(2 * i + 1) * 64];
            b.hi = __builtin_bit_cast(h8, f32x4{h[0], h[1], 0.0f, 0.0f});
            b.lo = __builtin_bit_cast(h8, f32x4{lo2[0], lo2[1], 0.0f, 0.0f});
        }
        return b;
    };
    // A operands of one work item (path, output tile): one (hi, lo) pair per K-chunk of the path's input degree, requested one item ahead
    const int oSv = opaque_s(Q.o_S_val);
    constexpr int NVI = val16_num_items<L>();
    struct VOps { f32x4 ah[2], al[2]; };
    auto load_vops = [&]<int I>() {
        VOps o{};
        if constexpr (I < NVI) {
            constexpr VItem16 it = val16_item<L>(I);
            constexpr int NC = feat_chunks16(dtp_path<L>(it.p).l1);
            static_for<NC>([&]<int c>() {
                constexpr int s = val16_slot<L>(it.p, c, it.To);
                o.ah[c] = wld4((oSv + s * 512) * 4); o.al[c] = wld4((oSv + s * 512 + 256) * 4);
            });
        }
        return o;
    };
    VOps vnxt = load_vops.template operator()<0>();
    static_for<L + 1>([&]<int l3>() {
        constexpr int d3 = 2 * l3 + 1, NTo = val_tiles16(l3);
        float val[d3][NTo][4];
        static_for<d3>([&]<int K>() { static_for<NTo>([&]<int To>() {
            if constexpr (l3 == 0) { const f32x4 bb = *reinterpret_cast<const f32x4*>(rows + RW::bval0 + 16 * To + 4 * g); static_for<4>([&]<int r>() { val[K][To][r] = bb[r]; }); }
            else static_for<4>([&]<int r>() { val[K][To][r] = 0.0f; });
        }); });
        constexpr int I0 = val16_group_first<L>(l3), I1 = val16_group_first<L>(l3 + 1);
        static_for<I1 - I0>([&]<int dI>() {
            constexpr int I = I0 + dI;
            constexpr VItem16 it = val16_item<L>(I);
            constexpr PathInfo pi = dtp_path<L>(it.p);
            constexpr int l1 = pi.l1, l2 = pi.l2, d1 = 2 * l1 + 1, NC = feat_chunks16(l1), To = it.To;
            const VOps cur = vnxt;
            vnxt = load_vops.template operator()<I + 1>();
            using Cg = CG<l1, l2, l3>;
            float m[Cg::NM];
            Cg::make(Y.template get<l2>(), m);
            f32x4 G[d1][NC];             // independent accumulators: one per (component, chunk); terms outermost
            HL b[d1][NC];
            static_for<NC>([&]<int c>() { static_for<d1>([&]<int i>() { b[i][c] = load_B.template operator()<l1, i, c>(); }); });
            static_for<NC>([&]<int c>() { static_for<d1>([&]<int i>() { G[i][c] = mfma16(__builtin_bit_cast(h8, cur.ah[c]), b[i][c].hi, f32x4{0, 0, 0, 0}); }); });
            static_for<NC>([&]<int c>() { static_for<d1>([&]<int i>() { G[i][c] = mfma16(__builtin_bit_cast(h8, cur.ah[c]), b[i][c].lo, G[i][c]); }); });
            static_for<NC>([&]<int c>() { static_for<d1>([&]<int i>() { G[i][c] = mfma16(__builtin_bit_cast(h8, cur.al[c]), b[i][c].hi, G[i][c]); }); });
            static_for<4>([&]<int r>() {
                float o[d3];
                static_for<d3>([&]<int K>() { o[K] = val[K][To][r]; });
                static_for<d1>([&]<int i>() {
                    float gv = G[i][0][r];
                    if constexpr (NC == 2) gv += G[i][1][r];
                    Cg::template acc<i>(gv, m, o);
                });
                static_for<d3>([&]<int K>() { opaque_v(o[K]); val[K][To][r] = o[K]; });
            });
            sched_fence();
        });
        // this degree's block of the segment records (internal layout [l][m][channel]); head of a channel = channel / (mul / 4)
        constexpr int NS = d3 * NTo;
        float x[NS][4]; int ro[NS]; float iv[NS];
        static_for<d3>([&]<int K>() { static_for<NTo>([&]<int To>() {
            constexpr int n = K * NTo + To;
            ro[n] = blk_off(l3) + K * mul_of(l3) + 16 * To + 4 * g;
            int hd;
            if constexpr (l3 == 0) hd = To; else if constexpr (l3 == 1) hd = 2 * To + (g >> 1); else hd = g;
            const float pwh = hd == 0 ? pw[0] : (hd == 1 ? pw[1] : (hd == 2 ? pw[2] : pw[3]));
            iv[n] = hd == 0 ? inv_s[0] : (hd == 1 ? inv_s[1] : (hd == 2 ? inv_s[2] : inv_s[3]));
            static_for<4>([&]<int r>() { x[n][r] = val[K][To][r] * cv[l3]; });
            if (drec != nullptr && valid) st4(drec + ro[n], f32x4{x[n][0], x[n][1], x[n][2], x[n][3]});
            static_for<4>([&]<int r>() { x[n][r] *= pwh; });
        }); });
        emit(x, ro, iv);
        sched_fence();
    });
    if (drec != nullptr && valid && g == 0) st4(drec + D, f32x4{logit[0], logit[1], logit[2], logit[3]});
}

// persistent waves striding over 16-edge tiles; tile table from the neighbour search (tile_info[16 ..]: edge prefix per scale)
template <int L> __global__ __launch_bounds__(64, 2) void k_edge16(Edge16Params Q) {
    __shared__ __attribute__((aligned(16))) float rows[Rows16<L>::total];
#if defined(DEDF_E16_LDS_TIMING)
    __shared__ __attribute__((aligned(16))) char park[Park16<L>::bytes + 1024];
    reinterpret_cast<f32x4*>(park + Park16<L>::bytes)[lane_id()] = f32x4{0, 0, 0, 0};
#else
    __shared__ __attribute__((aligned(16))) char park[Park16<L>::bytes];
#endif
    const EdgeParams& P = Q.E;
    const int lane = lane_id();
    const Buf wb = make_buf(Q.W, Q.W_bytes);
    {
        using RW = Rows16<L>;
        auto cp = [&](int dstf, int srcf, int n) { for (int i = lane; i < n; i += 64) rows[dstf + i] = Q.W[srcf + i]; };
        cp(RW::off3, Q.o_off3, dtp_wn<L>()); cp(RW::b0, Q.o_b0, lin0_tiles16<L>() * 16); cp(RW::bval0, Q.o_bval0, 64); cp(RW::adot, Q.o_adot, 64);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    const int* ti = P.tile_info;
    if (ti[40]) return;                                   // edge workspace overflow: nothing to do (the 32-edge table has no tiles then)
    int pre[kMaxScales + 1];
    pre[0] = 0;
    for (int n = 0; n < P.n_scales; ++n) pre[n + 1] = pre[n] + (ti[16 + n + 1] - ti[16 + n] + 15) / 16;
    const int ntiles = pre[P.n_scales];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int scale = 0;
        while (t >= pre[scale + 1]) ++scale;
        const int k = t - pre[scale];
        const int ebase = ti[16 + scale], En = ti[16 + scale + 1] - ebase;
        edge16_tile<L>(Q, wb, lane, scale, ebase + 16 * k, min(16, En - 16 * k), rows, park);
    }
}

}  // namespace dedf
