// Compile-time description of the score-head network for irreps  64x0e + 32x1e (+ 16x2e (+ 8x3e)),  SH 0..L,  L = 1, 2, 3.
// Everything here is constexpr and shared by the host weight packers and the device kernels, so the order in which a
// kernel walks K-steps and the order in which the host lays out the A operands cannot drift apart.
//
// Path / weight ordering restates the reference's instruction construction:
//   DepthwiseTensorProduct   equiformer/tensor_product_rescale.py:352-382  (creation order in1 -> in2 -> l_out,
//                            outputs re-sorted by l with a stable sort; weight blocks stay in creation order)
//   LinearRS / FCTP          equiformer/tensor_product_rescale.py:155-185
#pragma once
#include "dedf_layout.h"
#include "dedf_tables.h"

namespace dedf {

constexpr int kHeads = 4;
constexpr int kFc0 = 128, kFc1 = 128, kFc2 = 64;     // fc_neurons of every shipped config (resolved)
// radial table of the sampler (dedf_edge.h: EdgeParams::rtab): grid intervals of a finite scale over [0, r) and of the all-pairs scale over
// [0, kRtabInfiniteSpan * length_enc_max_r) (longer edges fall back to the per-edge evaluation)
// (round 4: the all-pairs grid 32 768 -> 16 384 intervals: its 4-point interpolation error stays far below the encoder's own fp32 argument noise --
//  guard word 5e-5 against 1.1e-4, table vs per-edge step 6.5e-6 against 5.4e-6 of the displacement -- and the generator fits one round of
//  workgroups: 705 tiles instead of 1 217; 8 192 intervals sit at the guard's bound: profiles/r04l_rtab_grid.log)
constexpr int kRtabFinite = 2048, kRtabInfinite = 16384;
constexpr double kRtabInfiniteSpan = 1.5;
constexpr int kRtabMinNodes = 8192;        // pose x query nodes below which the sampler evaluates the front per edge (small batches: the
                                           // table's generator launch costs more than it saves)
constexpr int kLenEmb = 64, kTimeEmb = 64, kTimeEnc = 256, kTimeHid = 128;
constexpr int kMaxScales = 8;

// Multiplicities.  The reference's irreps are 64x0e + 32x1e + 16x2e + 8x3e (true_mul).  The kernels work on whole 16-channel chunks
// (one K = 16 step of the fp16 MFMAs, 8 accumulator registers of each half-wave), so the 8x3e block runs as a ZERO-PADDED 16x3e block
// (mul_of): an exact embedding -- every operation is linear per channel, a gate / activation that maps 0 to 0, a LayerNorm whose
// statistics are taken over the true channels, or the per-head softmax -- with the true channels placed per head (pad_pos: the kernels'
// head of a channel is channel / (mul / 4), the reference's Vec2AttnHeads takes true_mul / 4 consecutive channels per head).  The host
// builds the zero-padded parameters (dedf_pack.h::pad_params); the C ABI speaks the true shapes.
DEDF_HD constexpr int mul_of(int l) { return l >= 3 ? 16 : 64 >> l; }
DEDF_HD constexpr int true_mul(int l) { return 64 >> l; }
DEDF_HD constexpr int pad_pos(int l, int c) { return (c / (true_mul(l) / 4)) * (mul_of(l) / 4) + c % (true_mul(l) / 4); }
// hidden multiplicities of the FFN (irreps_mlp_mid x mul): 192 / 96 / 48 / 24 -- the last one runs padded to one 32-row tile
DEDF_HD constexpr int hid_of(int l) { return l >= 3 ? 32 : 3 * (64 >> l); }
DEDF_HD constexpr int true_hid(int l) { return 3 * (64 >> l); }
DEDF_HD constexpr int iabs(int a) { return a < 0 ? -a : a; }
DEDF_HD constexpr int imin(int a, int b) { return a < b ? a : b; }
DEDF_HD constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
DEDF_HD constexpr int rup(int a, int b) { return cdiv(a, b) * b; }

template <int L> DEDF_HD constexpr int feat_dim() { int d = 0; for (int l = 0; l <= L; ++l) d += mul_of(l) * (2 * l + 1); return d; }      // 160 / 240 / 352 (kernel layout)
template <int L> DEDF_HD constexpr int true_feat_dim() { int d = 0; for (int l = 0; l <= L; ++l) d += true_mul(l) * (2 * l + 1); return d; }   // 160 / 240 / 296
DEDF_HD constexpr int true_blk_off(int l) { int d = 0; for (int i = 0; i < l; ++i) d += true_mul(i) * (2 * i + 1); return d; }
// offset of irreps block l (same in the reference layout [mul][m] and in the internal layout [m][mul])
DEDF_HD constexpr int blk_off(int l) { int d = 0; for (int i = 0; i < l; ++i) d += mul_of(i) * (2 * i + 1); return d; }
template <int L> DEDF_HD constexpr int sum_mul() { int d = 0; for (int l = 0; l <= L; ++l) d += mul_of(l); return d; }

struct PathInfo {
    int l1, l2, l3;
    int mul1, mul2;   // multiplicities of in1 / in2 (mul2 == 1 for the SH depth-wise TP)
    int wstart;       // offset of this path's weight block in the flat e3nn weight vector
    int kofs;         // offset of this path's channels inside the l3 block of the (sorted) DTP output
};

// ---- depth-wise TP  features (x) SH(0..L), outputs l3 <= L ------------------------------------------------------
template <int L> DEDF_HD constexpr int dtp_num_paths() {
    int n = 0;
    for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
        for (int l3 = iabs(l1 - l2); l3 <= imin(L, l1 + l2); ++l3) ++n;
    return n;
}
template <int L> DEDF_HD constexpr PathInfo dtp_path(int p) {
    int n = 0, w = 0;
    int kc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
        for (int l3 = iabs(l1 - l2); l3 <= imin(L, l1 + l2); ++l3) {
            if (n == p) return PathInfo{l1, l2, l3, mul_of(l1), 1, w, kc[l3]};
            w += mul_of(l1); kc[l3] += mul_of(l1); ++n;
        }
    return PathInfo{-1, -1, -1, 0, 0, w, 0};
}
template <int L> DEDF_HD constexpr int dtp_wn() { return dtp_path<L>(dtp_num_paths<L>()).wstart; }   // 480 (L=2), 224 (L=1)
template <int L> DEDF_HD constexpr int dtp_k(int l3) {       // channels of the l3 block of the DTP output: 112/192/176
    int k = 0;
    for (int p = 0; p < dtp_num_paths<L>(); ++p) if (dtp_path<L>(p).l3 == l3) k += dtp_path<L>(p).mul1;
    return k;
}
template <int L> DEDF_HD constexpr int dtp_out_dim() { int d = 0; for (int l = 0; l <= L; ++l) d += dtp_k<L>(l) * (2 * l + 1); return d; }
template <int L> DEDF_HD constexpr int dtp_path_of_row(int wrow) {
    for (int p = 0; p < dtp_num_paths<L>(); ++p) {
        const PathInfo pi = dtp_path<L>(p);
        if (wrow >= pi.wstart && wrow < pi.wstart + pi.mul1) return p;
    }
    return -1;
}
// A "group" = 8 consecutive weight rows = 4 accumulator registers x 2 half-waves = 4 MFMA K-steps.
// Index of w-group `wg` (= wrow / 8) among the groups that feed the same l3 (their K-steps follow w order).
template <int L> DEDF_HD constexpr int dtp_group_index(int wg) {
    const int l3 = dtp_path<L>(dtp_path_of_row<L>(wg * 8)).l3;
    int n = 0;
    for (int g = 0; g < wg; ++g) if (dtp_path<L>(dtp_path_of_row<L>(g * 8)).l3 == l3) ++n;
    return n;
}
// K-steps (as indices into the l3 block of the sorted DTP output) in the order the kernels walk them
template <int L> inline std::vector<KStep> dtp_steps(int l3) {
    std::vector<KStep> s;
    for (int wg = 0; wg < dtp_wn<L>() / 8; ++wg) {
        const PathInfo pi = dtp_path<L>(dtp_path_of_row<L>(wg * 8));
        if (pi.l3 != l3) continue;
        const int u0 = wg * 8 - pi.wstart;
        for (int j = 0; j < 4; ++j) s.push_back({pi.kofs + u0 + j, pi.kofs + u0 + 4 + j});
    }
    return s;
}

// A "chunk" = 16 consecutive channels of one depth-wise-TP path = 8 accumulator registers x 2 half-waves = one K = 16 step
// of the split-fp16 MFMAs.  The edge kernel walks the chunks grouped by output degree l3 (all l3 = 0 chunks, then l3 = 1,
// ...) so that only one group's accumulators are live at a time, and inside a group by (l1, channel range) so that
// consecutive chunks of different paths read the same input rows (dtp_pos_same_x); the host packs the last radial-MLP
// layer (rows = per-edge TP weights) and the A-operand streams in that walk order.
//   position p (0 .. WN/16)  ->  e3nn weight chunk dtp_pos_chunk(p)  (weight rows 16 wc .. 16 wc + 15)
// Element k16 = 8 h + jj of the chunk at position p (h = half-wave, jj = register 8 (p % 2) + jj of weight tile p / 2) is
// channel chunk_row(k16) of the chunk: the register order of a row-layout tile.
DEDF_HD constexpr int chunk_row(int k16) { const int h = k16 >> 3, jj = k16 & 7; return (jj & 3) + 8 * (jj >> 2) + 4 * h; }
template <int L> struct DtpWalk { int chunk[64]; PathInfo path[64]; int n; };
// A path whose input degree is lower than its output degree is evaluated in OUTPUT-SIDE form by the edge kernel's first depth-wise TP
// as well (see make_val_walk): B operands = per-edge weight x message component (2 l1 + 1 of them per channel instead of the 2 l3 + 1
// contracted ones), the contraction with the SH follows the GEMM.  Fewer operand splits, MFMAs and Clebsch-Gordan multiply-adds.
// (input degrees 0 and 1 only: one or three G tiles in flight)
template <int L> DEDF_HD constexpr bool dtp_path_out_side(const PathInfo& pi) { return pi.l1 < pi.l3 && pi.l1 <= 1; }
template <int L> DEDF_HD constexpr DtpWalk<L> make_dtp_walk() {
    DtpWalk<L> w{};
    int n = 0;
    for (int l3 = 0; l3 <= L; ++l3) {
        // input-side paths: by input degree and channel range, so that consecutive chunks of different paths share their source rows
        for (int l1 = 0; l1 <= L; ++l1)
            for (int c = 0; c < mul_of(l1) / 16; ++c)
                for (int q = 0; q < dtp_num_paths<L>(); ++q) {
                    const PathInfo pi = dtp_path<L>(q);
                    if (pi.l3 != l3 || pi.l1 != l1 || dtp_path_out_side<L>(pi)) continue;
                    w.chunk[n] = pi.wstart / 16 + c; w.path[n] = pi; ++n;
                }
        // output-side paths last (their contraction then meets the group's activations), every path's chunks contiguous
        for (int q = 0; q < dtp_num_paths<L>(); ++q) {
            const PathInfo pi = dtp_path<L>(q);
            if (pi.l3 != l3 || !dtp_path_out_side<L>(pi)) continue;
            for (int c = 0; c < mul_of(pi.l1) / 16; ++c) { w.chunk[n] = pi.wstart / 16 + c; w.path[n] = pi; ++n; }
        }
    }
    w.n = n;
    for (int i = n; i < 64; ++i) { w.chunk[i] = dtp_wn<L>() / 16; w.path[i] = PathInfo{-1, -1, -1, 0, 0, 0, 0}; }
    return w;
}
// Edge-aligned-frame ("SO(2)") form of the first depth-wise TP (S = true; diffusion_edf_amd/so2.py, dedf_tables.h::kSo2*): the source rows are
// rotated into the frame whose polar axis is the edge, where every path reaches output component k from ONE source component (|m| equal) --
// B operand = per-edge weight x rotated component, no Clebsch-Gordan sum, no output-side paths.  The scalar outputs (l3 = 0) are frame
// independent and keep the general form; ALL chunks with l3 >= 1 form one group walked by (input degree, channel range), so that a chunk of
// source rows is rotated once and serves every path that reads it (five at l1 = 1, 2 of lmax 2).  The l3 >= 1 accumulators are live together.
// (lmax 3: two such groups, l3 in {1, 2} and l3 = 3 -- all three degrees at once are ten paired accumulator tiles, more than the stage has
//  registers for; the source rows are then rotated once per group.)
template <int L> DEDF_HD constexpr int so2_num_groups() { return L >= 3 ? 3 : 2; }
template <int L> DEDF_HD constexpr int so2_group_of(int l3) { return l3 == 0 ? 0 : (L >= 3 && l3 == 3 ? 2 : 1); }
template <int L> DEDF_HD constexpr DtpWalk<L> make_dtp_walk_so2() {
    DtpWalk<L> w{};
    int n = 0;
    for (int g = 0; g < so2_num_groups<L>(); ++g)
        for (int l1 = 0; l1 <= L; ++l1)
            for (int c = 0; c < mul_of(l1) / 16; ++c)
                for (int q = 0; q < dtp_num_paths<L>(); ++q) {
                    const PathInfo pi = dtp_path<L>(q);
                    if (so2_group_of<L>(pi.l3) != g || pi.l1 != l1) continue;
                    w.chunk[n] = pi.wstart / 16 + c; w.path[n] = pi; ++n;
                }
    w.n = n;
    for (int i = n; i < 64; ++i) { w.chunk[i] = dtp_wn<L>() / 16; w.path[i] = PathInfo{-1, -1, -1, 0, 0, 0, 0}; }
    return w;
}
template <int L, bool S = false> inline constexpr DtpWalk<L> kDtpWalk = S ? make_dtp_walk_so2<L>() : make_dtp_walk<L>();
template <int L, bool S = false> DEDF_HD constexpr int dtp_pos_chunk(int p) { return p < 64 ? kDtpWalk<L, S>.chunk[p] : dtp_wn<L>() / 16; }
template <int L, bool S = false> DEDF_HD constexpr PathInfo dtp_pos_path(int p) { return p < 64 ? kDtpWalk<L, S>.path[p] : PathInfo{-1, -1, -1, 0, 0, 0, 0}; }
template <int L, bool S = false> DEDF_HD constexpr int dtp_pos_l3(int p) { return dtp_pos_path<L, S>(p).l3; }
// first channel of the chunk inside its path (u0) / inside the l3 block of the sorted DTP output
template <int L, bool S = false> DEDF_HD constexpr int dtp_pos_u0(int p) { return dtp_pos_chunk<L, S>(p) * 16 - dtp_pos_path<L, S>(p).wstart; }
template <int L, bool S = false> DEDF_HD constexpr int dtp_pos_channel(int p, int k16) { return dtp_pos_path<L, S>(p).kofs + dtp_pos_u0<L, S>(p) + chunk_row(k16); }
// chunks p and q read the same input channels (same l1, same channel range)
template <int L, bool S = false> DEDF_HD constexpr bool dtp_pos_same_x(int p, int q) {
    return p >= 0 && q >= 0 && p < dtp_wn<L>() / 16 && q < dtp_wn<L>() / 16 && dtp_pos_path<L, S>(p).l1 == dtp_pos_path<L, S>(q).l1 && dtp_pos_u0<L, S>(p) == dtp_pos_u0<L, S>(q);
}
// output-side chunk at walk position p?  first / last chunk of its path?
template <int L, bool S = false> DEDF_HD constexpr bool dtp_pos_out(int p) { return !S && p >= 0 && p < dtp_wn<L>() / 16 && dtp_path_out_side<L>(dtp_pos_path<L, S>(p)); }
template <int L, bool S = false> DEDF_HD constexpr bool dtp_pos_path_first(int p) { return dtp_pos_u0<L, S>(p) == 0; }
template <int L, bool S = false> DEDF_HD constexpr bool dtp_pos_path_last(int p) { return dtp_pos_u0<L, S>(p) + 16 == dtp_pos_path<L, S>(p).mul1; }
// the output-side path ending at p is the first of its output degree to be contracted (the VALU-side accumulators start there)
template <int L> DEDF_HD constexpr bool dtp_pos_opens_vacc(int p) {
    if (!dtp_pos_out<L>(p) || !dtp_pos_path_last<L>(p)) return false;
    for (int q = 0; q < p; ++q)
        if (dtp_pos_out<L>(q) && dtp_pos_path_last<L>(q) && dtp_pos_l3<L>(q) == dtp_pos_l3<L>(p)) return false;
    return true;
}
template <int L, bool S = false> DEDF_HD constexpr bool dtp_group_has_out(int l3) {
    for (int q = 0; q < dtp_wn<L>() / 16; ++q) if (dtp_pos_out<L, S>(q) && dtp_pos_l3<L, S>(q) == l3) return true;
    return false;
}
// e3nn weight row held by row r of the (walk-ordered) last radial layer
template <int L, bool S = false> DEDF_HD constexpr int dtp_walk_row(int r) { return dtp_pos_chunk<L, S>(r / 16) * 16 + r % 16; }
// number of chunks with output degree <= l3 (= position one past the end of group l3)
template <int L> DEDF_HD constexpr int dtp_group_end(int l3) { return (dtp_k<L>(0) + (l3 >= 1 ? dtp_k<L>(1) : 0) + (l3 >= 2 ? dtp_k<L>(2) : 0) + (l3 >= 3 ? dtp_k<L>(3) : 0)) / 16; }
// A-operand stream of a linear layer fed by the depth-wise TP: chunks in walk order; an l3 = 0 chunk feeds nt0 output
// tiles (one 512-float slot each, consumed in items of two tiles), an l3 >= 1 chunk one tile.  Slot = hi image | lo image.
struct DtpItem { int pos, t, slot, ntile; };      // chunk position, item inside the chunk, first slot, tiles in the item
template <int L, bool S = false> DEDF_HD constexpr DtpItem dtp_item(int I, int nt0) {
    int i = 0, slot = 0;
    for (int p = 0; p < dtp_wn<L>() / 16; ++p) {
        const bool z = dtp_pos_l3<L, S>(p) == 0;
        const int ni = z ? cdiv(nt0, 2) : 1;
        if (I < i + ni) { const int t = I - i; return DtpItem{p, t, slot + 2 * t, z ? imin(2, nt0 - 2 * t) : 1}; }
        i += ni; slot += z ? nt0 : 1;
    }
    return DtpItem{dtp_wn<L>() / 16, 0, slot, 0};
}
template <int L, bool S = false> DEDF_HD constexpr int dtp_item_first(int p, int nt0) {
    int i = 0;
    for (int c = 0; c < p; ++c) i += dtp_pos_l3<L, S>(c) == 0 ? cdiv(nt0, 2) : 1;
    return i;
}
template <int L, bool S = false> DEDF_HD constexpr int dtp_num_slots(int nt0) { return dtp_item<L, S>(1 << 20, nt0).slot; }
// (S: one A slot per chunk as in the general form -- the path's reference coefficient kSo2Ref is folded into the slot by the host packer, the
//  kernel multiplies the per-edge weights by the compile-time ratios of the other terms)
template <int L> DEDF_HD constexpr float so2_ref(const PathInfo& pi) { return kSo2Ref[pi.l1][pi.l2][pi.l3]; }
// walk position one past the end of edge-frame group g
template <int L> DEDF_HD constexpr int so2_group_end(int g) {
    int n = 0;
    for (int p = 0; p < dtp_wn<L>() / 16; ++p) if (so2_group_of<L>(kDtpWalk<L, true>.path[p].l3) <= g) ++n;
    return n;
}

// ---- second depth-wise TP (attention value) in OUTPUT-SIDE form -----------------------------------------------------------
// value[o,k] = sum_p sum_u W2[p,u,o] sum_ij C^p_ijk u[u,i] Y[j]  is evaluated as  G^p_i[o] = sum_u W2[p,u,o] u[u,i]  (a GEMM whose
// B operand is component i of the gated features themselves -- split into fp16 hi / lo ONCE when they are gated and parked in LDS,
// 240 values per edge instead of the 1568 depth-wise-TP outputs) followed by the lane-local contraction
// value[o,k] += (sum_j C_ijk Y[j]) G_i[o].  Same FLOPs, the Clebsch-Gordan work moves behind the GEMM.
// Parked B chunks (16 channels = 8 registers of each half-wave, dedf_layout.h::chain_k): slot q of (degree l, component i, chunk c).
template <int L> DEDF_HD constexpr int park_slot(int l, int i, int c) {
    int q = 0;
    for (int a = 0; a < l; ++a) q += (2 * a + 1) * (mul_of(a) / 16);
    return q + i * (mul_of(l) / 16) + c;
}
template <int L> DEDF_HD constexpr int park_slots() { return park_slot<L>(L + 1, 0, 0); }          // 15 (L = 2), 10 (L = 1)
// Zero padding inside a 16-channel block (lmax 3: 8x3e as 16x3e, true channels at pad_pos = the first two of every group of four): of the 8
// registers that hold a 16-row block in either half-wave, and of the 4 channels of a lane's run of source rows, those with index % 4 >= 2 are
// padding -- exactly 0 in every operand and result (and in every narrower UNet level, whose true channels are a subset).  The lane-local
// Clebsch-Gordan work skips them.
// NW ("narrow"): a UNet layer whose source AND destination irreps are the narrow level shape 32x0e+16x1e+8x2e(+4x3e) (levels 0-1 of the panda
// UNets, unet_feature_extractor.py:141-202 with configs/panda_mug/pick_lowres/score_model_configs.yaml:33-55) embedded in the kernel shapes with
// its true channels at the positions p with p % 4 < pad_live (diffusion_edf_amd/unet_pad.py::place): HALF of every block of degree <= 2 and a
// QUARTER of the l = 3 block are structural zeros, and the narrow instantiations (k_edge<..., NW = true>) skip the lane-local work on them
// like the lmax-3 kernels skip the padding of 8x3e.  The matrix products still run on the padded shapes (K = 16 steps, 32-row tiles).
template <int L, bool NW = false> DEDF_HD constexpr int pad_live(int l) { return NW ? (l >= 3 ? 1 : 2) : ((L == 3 && l == 3) ? 2 : 4); }
template <int L, bool NW = false> DEDF_HD constexpr bool pad_reg(int l, int r) { return (r % 4) >= pad_live<L, NW>(l); }
// of the first n registers of a run that starts at a multiple of four: how many hold true channels / the index of register r among them
template <int L, bool NW> DEDF_HD constexpr int live_count(int l, int n) { int c = 0; for (int r = 0; r < n; ++r) c += pad_reg<L, NW>(l, r) ? 0 : 1; return c; }
// The l3 >= 2 accumulators of the first depth-wise TP's linear at lmax 3 hold two components per 32-row tile (dedf_edge.h::mfma_chunk):
// 48 + 64 instead of 80 + 112 accumulator registers in the groups where k_edge<3> spills.
#ifndef DEDF_PAIR_L2
#define DEDF_PAIR_L2 0      // measured at lmax 2: see DESIGN.md section 5.0
#endif
template <int L> DEDF_HD constexpr bool acc_paired(int l3) { return (L == 3 || (L == 2 && DEDF_PAIR_L2)) && l3 >= 2; }
// LDS slots (16 bytes per lane) of parked chunk q: the hi halves in slot park_phys(q), the residuals in the next one -- except the l = 3
// chunks at lmax 3: four of a lane's eight registers are the zero padding of 8x3e (pad_pos puts a head's two true channels first in its group
// of four: registers 0, 1, 4, 5 of either half-wave), so hi and lo of the four real ones share ONE slot (dedf_dev.h::split4pk).  7 KB less per
// wave, which is part of what lets four lmax-3 waves share a CU's 160 KB.
template <int L> DEDF_HD constexpr bool park_packed(int q) { return L == 3 && q >= park_slot<L>(3, 0, 0); }
template <int L> DEDF_HD constexpr int park_phys(int q) {
    if (L == 3 && q >= park_slot<L>(3, 0, 0)) return 2 * park_slot<L>(3, 0, 0) + (q - park_slot<L>(3, 0, 0));
    return 2 * q;
}
template <int L> DEDF_HD constexpr int park_phys_slots() { return park_phys<L>(park_slots<L>()); }      // 30 (L = 2), 20 (L = 1), 37 (L = 3)
// Work item of the value GEMMs: up to three independent accumulator tiles (consecutive MFMAs never hit the same one):
//   l1 = 0 : one component, two K-chunks at a time (partial sums, merged before the contraction)
//   l1 = 1 : the three components of one K-chunk          l1 = 2 : components {0,1,2}, then {3,4}, of the only K-chunk
//   l1 = 3 : components {0,1,2}, {3,4}, {5,6}
// A operands: one 512-float slot (hi image | lo image) per (path, output tile, K-chunk), in the order first used.
struct VItem {
    int p, t;              // path (dtp_path), output row tile inside the l3 block
    int na;                // accumulators
    int comp[3], bq[3], aslot[3];
    bool first, last;      // first / last item of its (path, tile, component set): zero the accumulators / contract them
    bool merge;            // l1 = 0: accumulators 0 and 1 are partial sums of component 0
    int group_end;         // output degree completed by this item's contraction, or -1
    int new_slots;         // A slots this item is the first to use (they follow the previous item's in the stream)
    bool in_side;          // INPUT-side item (val_path_in_side): comp[] are OUTPUT components, the B operands are contracted features formed on the fly
    bool chain;            // chained items (DEDF_VAL_CHAIN0 / 1): the accumulators ARE output tiles / components and run on from item to item through every
                           // chained path of the group; they are added to the value once, by the group's last item
    bool share_b;          // chained l3 = 0 items: accumulator a = output tile a, ONE B operand for all of them
};
// lmax 3, the six value paths from l1 = 3 into l3 <= 2 in INPUT-side form: the seven parked components of the 8x3e block (4 true channels per
// lane) are contracted with the SH first, B_k[u] = sum_ij C_ijk u[u,i] Y[j], and the GEMM yields the 2 l3 + 1 output components directly:
// 1 / 3 / 5 accumulator tiles per path instead of 7, and the Clebsch-Gordan multiply-adds run on 4 registers per lane instead of the 32 / 16 / 8
// that hold the output rows.  Same A operands in the same order (one slot per (path, tile, K-chunk)): the host packing does not change.
#ifndef DEDF_VAL_INSIDE3
#define DEDF_VAL_INSIDE3 1
#endif
// The scalar outputs (l3 = 0; paths (l, l, 0)) in CHAINED input-side form at every lmax: value0[o] = sum_l sum_u W2[(l,l,0),u,o] B_l[u] with
// B_0 = the parked scalars as they are and B_l[u] = sum_i C_ii0 u[u,i] Y_l[i] -- ONE contracted operand per K-chunk whatever the degree, so the
// group is a single GEMM over all its K-chunks into the two output tiles: 14 instead of 30 MFMA triples at lmax 2, no accumulator is read before the
// group ends, and the Clebsch-Gordan multiply-adds run on the 8 registers of a K-chunk instead of the 32 that hold the output rows.
#ifndef DEDF_VAL_CHAIN0
#define DEDF_VAL_CHAIN0 1
#endif
// The vector outputs (l3 = 1) of the paths where the input side is the narrow one, chained the same way behind the group's output-side paths:
// (1, 0, 1) -- its B operands are the parked components themselves, no VALU work at all -- and every path from l1 >= 2 (8 registers of a K-chunk
// against the 16 that hold the output rows, 3 accumulator tiles instead of 5 / 7).
#ifndef DEDF_VAL_CHAIN1
#define DEDF_VAL_CHAIN1 1
#endif
DEDF_HD constexpr bool val_path_chain1(const PathInfo& pi) { return DEDF_VAL_CHAIN1 && pi.l3 == 1 && ((pi.l1 == 1 && pi.l2 == 0) || pi.l1 >= 2); }
template <int L> DEDF_HD constexpr bool val_path_in_side(const PathInfo& pi) { return DEDF_VAL_INSIDE3 && L == 3 && pi.l1 == 3 && pi.l3 <= 2; }
template <int L> DEDF_HD constexpr int val_tiles(int l3) { return l3 == 0 ? mul_of(0) / 32 : 1; }
template <int L> struct ValWalk { VItem item[96]; int n, n_slots; };
template <int L> DEDF_HD constexpr ValWalk<L> make_val_walk() {
    ValWalk<L> w{};
    int n = 0, slot = 0;
    for (int l3 = 0; l3 <= L; ++l3) {
        int last_of_group = -1;
        bool chain_open = false;
        for (int pass = 0; pass < 2; ++pass)
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            const PathInfo pi = dtp_path<L>(p);
            if (pi.l3 != l3 || val_path_chain1(pi) != (pass == 1)) continue;
            const int kc = mul_of(pi.l1) / 16, d1 = 2 * pi.l1 + 1;
            if (val_path_chain1(pi)) {
                for (int c = 0; c < kc; ++c) {
                    VItem it{};
                    it.p = p; it.t = 0; it.na = 3; it.merge = false; it.group_end = -1; it.chain = true; it.in_side = !(pi.l1 == 1 && pi.l2 == 0);
                    for (int a = 0; a < 3; ++a) { it.comp[a] = a; it.bq[a] = park_slot<L>(pi.l1, it.in_side ? 0 : a, c); it.aslot[a] = slot; }
                    it.first = !chain_open; it.last = false; it.new_slots = 1;
                    chain_open = true;
                    slot += 1;
                    w.item[n++] = it;
                }
                last_of_group = n - 1;
                continue;
            }
            if (DEDF_VAL_CHAIN0 && l3 == 0) {
                for (int c = 0; c < kc; ++c) {
                    VItem it{};
                    it.p = p; it.t = 0; it.na = val_tiles<L>(0); it.merge = false; it.group_end = -1; it.chain = true; it.share_b = true; it.in_side = pi.l1 > 0;
                    for (int a = 0; a < it.na; ++a) { it.comp[a] = 0; it.bq[a] = park_slot<L>(pi.l1, 0, c); it.aslot[a] = slot + a; }
                    it.first = last_of_group < 0 && c == 0; it.last = false; it.new_slots = it.na;
                    slot += it.na;
                    w.item[n++] = it;
                }
                last_of_group = n - 1;
                continue;
            }
            for (int t = 0; t < val_tiles<L>(l3); ++t) {
                if (pi.l1 == 0) {
                    for (int c = 0; c < kc; c += 2) {
                        VItem it{};
                        it.p = p; it.t = t; it.na = 2; it.merge = true; it.group_end = -1;
                        for (int a = 0; a < 2; ++a) { it.comp[a] = 0; it.bq[a] = park_slot<L>(0, 0, c + a); it.aslot[a] = slot + a; }
                        it.first = c == 0; it.last = c + 2 >= kc; it.new_slots = 2;
                        slot += 2;
                        w.item[n++] = it;
                    }
                } else if (val_path_in_side<L>(pi)) {
                    for (int c = 0; c < kc; ++c) {
                        for (int k0 = 0; k0 < 2 * l3 + 1; k0 += 3) {
                            VItem it{};
                            it.p = p; it.t = t; it.merge = false; it.group_end = -1; it.in_side = true;
                            it.na = imin(3, 2 * l3 + 1 - k0);
                            for (int a = 0; a < it.na; ++a) { it.comp[a] = k0 + a; it.bq[a] = park_slot<L>(pi.l1, 0, c); it.aslot[a] = slot; }
                            it.first = c == 0; it.last = c == kc - 1; it.new_slots = k0 == 0 ? 1 : 0;
                            w.item[n++] = it;
                        }
                        slot += 1;
                    }
                } else {
                    for (int c = 0; c < kc; ++c) {
                        for (int i0 = 0; i0 < d1; i0 += 3) {
                            VItem it{};
                            it.p = p; it.t = t; it.merge = false; it.group_end = -1;
                            it.na = imin(3, d1 - i0);
                            for (int a = 0; a < it.na; ++a) { it.comp[a] = i0 + a; it.bq[a] = park_slot<L>(pi.l1, i0 + a, c); it.aslot[a] = slot; }
                            it.first = c == 0; it.last = c == kc - 1; it.new_slots = i0 == 0 ? 1 : 0;
                            w.item[n++] = it;
                        }
                        slot += 1;
                    }
                }
            }
            last_of_group = n - 1;
        }
        if (last_of_group >= 0) { w.item[last_of_group].group_end = l3; if (w.item[last_of_group].chain) w.item[last_of_group].last = true; }
    }
    w.n = n; w.n_slots = slot;
    return w;
}
template <int L> inline constexpr ValWalk<L> kValWalk = make_val_walk<L>();
template <int L> DEDF_HD constexpr int val_num_items() { return kValWalk<L>.n; }
template <int L> DEDF_HD constexpr int val_num_slots() { return kValWalk<L>.n_slots; }
template <int L> DEDF_HD constexpr VItem val_item(int I) { return I >= 0 && I < kValWalk<L>.n ? kValWalk<L>.item[I] : VItem{}; }
// output degree of the path A slot S belongs to (operands of 16-row blocks are fetched by half the lanes)
template <int L> DEDF_HD constexpr int val_slot_l3(int S) {
    for (int i = 0; i < kValWalk<L>.n; ++i)
        for (int a = 0; a < kValWalk<L>.item[i].na; ++a)
            if (kValWalk<L>.item[i].aslot[a] == S) return dtp_path<L>(kValWalk<L>.item[i].p).l3;
    return 0;
}
// item I completes the first contraction into its output degree (l3 >= 1: the value accumulators start from zero there)
template <int L> DEDF_HD constexpr bool val_item_opens_group(int I) {
    const int l3 = dtp_path<L>(kValWalk<L>.item[I].p).l3;
    for (int i = 0; i < I; ++i)
        if (kValWalk<L>.item[i].last && dtp_path<L>(kValWalk<L>.item[i].p).l3 == l3) return false;
    return kValWalk<L>.item[I].last;
}
// first A slot that item I is the first to use (= slots consumed before it)
template <int L> DEDF_HD constexpr int val_item_slot0(int I) {
    int s = 0;
    for (int i = 0; i < I && i < kValWalk<L>.n; ++i) s += kValWalk<L>.item[i].new_slots;
    return s;
}

// ---- second depth-wise TP (attention value) in the EDGE FRAME (S = true; see make_dtp_walk_so2) -----------------------------------------------
// The gated features are parked in the edge frame, where  value'[o, k] = sum_p sum_u (c^p_k W2[p, u, o]) u'[u, i_p(k)]:  every term is a plain GEMM
// whose B operand is a parked component AS IT IS and whose A operand carries the term's coefficient (folded by the host: one A slot per
// (path, K-chunk, coefficient class); terms whose coefficient is minus the class's take the B operand with its sign bits flipped).  No
// Clebsch-Gordan work, no operand forming, no accumulator is read before its output degree is complete.  Paths with l2 >= 1 carry the per-edge
// cut-off factor of the non-scalar SH (graph_parser.py:194-198), which cannot be folded: they accumulate into a second set of tiles,
// value' = set0 + cns * set1.  The value leaves the edge frame (Rot<l3>::out) when its degree is complete.
struct SItem {
    int p, c;              // path (dtp_path), K-chunk of its input block
    int l3, set;           // output degree; accumulator set (0: l2 = 0, 1: l2 >= 1)
    int na;                // MFMA triples
    int acc[7];            // output component k (l3 >= 1) / output tile (l3 = 0)
    int tile[7];           // accumulator tile inside (l3, set): acc, or acc / 2 where two components share a tile (acc_paired: 16-row outputs at lmax 3)
    bool up[7];            // paired tiles: the component lives in rows 16-31 (the A operand is fetched by the upper half of the lanes)
    int bq[7];             // parked chunk of the B operand (park_slot)
    bool neg[7];           // the B operand enters negated
    int aslot[7];
    bool first[7];         // first accumulation into that tile
    bool first_one[7];     // ... when both sets share one tile (tiles whose edges all have cut-off factor 1: dedf_edge.h, value stage, ONE)
    int new_slots;
    int group_end;         // output degree completed by this item, or -1
    float coef;            // what the host folds into this item's A slot(s)
};
// operands one item may carry (they are all in registers while the item runs): a coefficient class with more terms is split into items that
// share its A slot
template <int L> DEDF_HD constexpr int sval_max_ops() { return L >= 3 ? 4 : 5; }
template <int L> struct SValWalk { SItem item[128]; int n, n_slots; };
template <int L> DEDF_HD constexpr SValWalk<L> make_sval_walk() {
    SValWalk<L> w{};
    int n = 0, slot = 0;
    // Output degrees from the highest down to the scalars: the rotation back and the segmented reduction of a completed degree (dedf_edge.h:
    // finish_value / store_group, VALU only) run under the GEMMs of the next one, and what is left exposed at the end is the cheapest degree.
    for (int gi = 0; gi <= L; ++gi) {
        const int l3 = gi < L ? L - gi : 0;
        const bool PR = l3 >= 1 && acc_paired<L>(l3);
        bool seen[2][7] = {{false, false, false, false, false, false, false}, {false, false, false, false, false, false, false}};
        bool seen_one[7] = {false, false, false, false, false, false, false};
        for (int set = 0; set < 2; ++set)
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            const PathInfo pi = dtp_path<L>(p);
            if (pi.l3 != l3 || (pi.l2 > 0) != (set == 1)) continue;
            const int nt = kSo2NT[pi.l1][pi.l2][pi.l3];
            for (int c = 0; c < mul_of(pi.l1) / 16; ++c) {
                if (l3 == 0) {      // one term (k = 0 from the source's m = 0 component), val_tiles output tiles with their own A slots
                    SItem it{};
                    it.p = p; it.c = c; it.l3 = 0; it.set = set; it.na = val_tiles<L>(0); it.group_end = -1; it.coef = kSo2C[pi.l1][pi.l2][0][0];
                    for (int a = 0; a < it.na; ++a) {
                        it.acc[a] = a; it.tile[a] = a; it.up[a] = false; it.bq[a] = park_slot<L>(pi.l1, kSo2I[pi.l1][pi.l2][0][0], c); it.neg[a] = false; it.aslot[a] = slot + a;
                        it.first[a] = !seen[set][a]; seen[set][a] = true;
                        it.first_one[a] = !seen_one[a]; seen_one[a] = true;
                    }
                    it.new_slots = it.na; slot += it.na;
                    w.item[n++] = it;
                    continue;
                }
                bool done[7] = {false, false, false, false, false, false, false};
                for (int t0 = 0; t0 < nt; ++t0) {      // coefficient classes: terms with |c| equal share the A slot
                    if (done[t0]) continue;
                    const float c0 = kSo2C[pi.l1][pi.l2][pi.l3][t0];
                    SItem it{};
                    it.p = p; it.c = c; it.l3 = l3; it.set = set; it.na = 0; it.group_end = -1; it.coef = c0; it.new_slots = 1;
                    for (int t = t0; t < nt; ++t) {
                        const float ct = kSo2C[pi.l1][pi.l2][pi.l3][t];
                        if (done[t] || !(ct == c0 || ct == -c0)) continue;
                        if (it.na == sval_max_ops<L>()) {      // the class goes on in another item on the same A slot
                            w.item[n++] = it;
                            it.na = 0; it.new_slots = 0;
                        }
                        done[t] = true;
                        const int a = it.na++, k = kSo2K[pi.l1][pi.l2][pi.l3][t], tl = PR ? k / 2 : k;
                        it.acc[a] = k; it.tile[a] = tl; it.up[a] = PR && (k % 2 == 1); it.bq[a] = park_slot<L>(pi.l1, kSo2I[pi.l1][pi.l2][pi.l3][t], c); it.neg[a] = ct != c0; it.aslot[a] = slot;
                        it.first[a] = !seen[set][tl]; seen[set][tl] = true;
                        it.first_one[a] = !seen_one[tl]; seen_one[tl] = true;
                    }
                    slot += 1;
                    w.item[n++] = it;
                }
            }
        }
        w.item[n - 1].group_end = l3;
    }
    w.n = n; w.n_slots = slot;
    return w;
}
template <int L> inline constexpr SValWalk<L> kSValWalk = make_sval_walk<L>();
template <int L> DEDF_HD constexpr int sval_num_items() { return kSValWalk<L>.n; }
template <int L> DEDF_HD constexpr int sval_num_slots() { return kSValWalk<L>.n_slots; }
template <int L> DEDF_HD constexpr SItem sval_item(int I) { return I >= 0 && I < kSValWalk<L>.n ? kSValWalk<L>.item[I] : SItem{}; }
template <int L> DEDF_HD constexpr int sval_slot_l3(int S) {
    for (int i = 0; i < kSValWalk<L>.n; ++i)
        for (int a = 0; a < kSValWalk<L>.item[i].na; ++a)
            if (kSValWalk<L>.item[i].aslot[a] == S) return kSValWalk<L>.item[i].l3;
    return 0;
}
// last item of output degree l3 in the walk
template <int L> DEDF_HD constexpr int sval_group_last(int l3) {
    for (int i = 0; i < kSValWalk<L>.n; ++i) if (kSValWalk<L>.item[i].group_end == l3) return i;
    return -1;
}
template <int L> DEDF_HD constexpr int sval_item_slot0(int I) {
    int s = 0;
    for (int i = 0; i < I && i < kSValWalk<L>.n; ++i) s += kSValWalk<L>.item[i].new_slots;
    return s;
}
// component k (output tile for l3 = 0) of set `set` of output degree l3 receives anything at all
template <int L> DEDF_HD constexpr bool sval_acc_used(int l3, int set, int a) {
    for (int i = 0; i < kSValWalk<L>.n; ++i) {
        const SItem& it = kSValWalk<L>.item[i];
        if (it.l3 != l3 || it.set != set) continue;
        for (int q = 0; q < it.na; ++q) if (it.acc[q] == a) return true;
    }
    return false;
}
// slot S is used by an operand that lives in rows 0-15 / in rows 16-31 of a paired tile
template <int L> DEDF_HD constexpr bool sval_slot_needs(int S, bool up) {
    for (int i = 0; i < kSValWalk<L>.n; ++i)
        for (int a = 0; a < kSValWalk<L>.item[i].na; ++a)
            if (kSValWalk<L>.item[i].aslot[a] == S && kSValWalk<L>.item[i].up[a] == up) return true;
    return false;
}

// ---- score tensor products: in1 = rotated query feature, in2 = field, outputs l3 in {0, 1} ------------------------
template <int L> DEDF_HD constexpr int stp_num_paths() {
    int n = 0;
    for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
        for (int l3 = iabs(l1 - l2); l3 <= imin(1, l1 + l2); ++l3) ++n;
    return n;
}
template <int L> DEDF_HD constexpr PathInfo stp_path(int p) {
    int n = 0, w = 0;
    int kc[2] = {0, 0};
    for (int l1 = 0; l1 <= L; ++l1) for (int l2 = 0; l2 <= L; ++l2)
        for (int l3 = iabs(l1 - l2); l3 <= imin(1, l1 + l2); ++l3) {
            if (n == p) return PathInfo{l1, l2, l3, mul_of(l1), mul_of(l2), w, kc[l3]};
            w += mul_of(l1) * mul_of(l2); kc[l3] += mul_of(l1); ++n;
        }
    return PathInfo{-1, -1, -1, 0, 0, w, 0};
}
template <int L> DEDF_HD constexpr int stp_wn() { return stp_path<L>(stp_num_paths<L>()).wstart; }    // 11776 (L=2)
template <int L> DEDF_HD constexpr int stp_k(int l3) {
    int k = 0;
    for (int p = 0; p < stp_num_paths<L>(); ++p) if (stp_path<L>(p).l3 == l3) k += stp_path<L>(p).mul1;
    return k;
}
// group index (4 K-steps each) of (path p, u-group gu) among the groups feeding l3, walking paths in creation order
template <int L> DEDF_HD constexpr int stp_group_index(int p, int gu) {
    const int l3 = stp_path<L>(p).l3;
    int n = 0;
    for (int q = 0; q < p; ++q) if (stp_path<L>(q).l3 == l3) n += stp_path<L>(q).mul1 / 8;
    return n + gu;
}
// chunk index (16 channels each) of (path p, chunk cu) among the chunks feeding l3, walking paths in creation order
template <int L> DEDF_HD constexpr int stp_chunk_index(int p, int cu) {
    const int l3 = stp_path<L>(p).l3;
    int n = 0;
    for (int q = 0; q < p; ++q) if (stp_path<L>(q).l3 == l3) n += stp_path<L>(q).mul1 / 16;
    return n + cu;
}
template <int L> inline std::vector<KStep> stp_steps(int l3) {
    std::vector<KStep> s;
    for (int p = 0; p < stp_num_paths<L>(); ++p) {
        const PathInfo pi = stp_path<L>(p);
        if (pi.l3 != l3) continue;
        for (int gu = 0; gu < pi.mul1 / 8; ++gu)
            for (int j = 0; j < 4; ++j) s.push_back({pi.kofs + gu * 8 + j, pi.kofs + gu * 8 + 4 + j});
    }
    return s;
}
// same walk for the second depth-wise TP of the attention value (shared weights; paths in creation order)
template <int L> DEDF_HD constexpr int dtp2_group_index(int p, int gu) {
    const int l3 = dtp_path<L>(p).l3;
    int n = 0;
    for (int q = 0; q < p; ++q) if (dtp_path<L>(q).l3 == l3) n += dtp_path<L>(q).mul1 / 8;
    return n + gu;
}

// ---- row spaces of the edge kernel ---------------------------------------------------------------------------------
// l3 = 0 GEMM of sep_act.lin and sep_alpha share their K-steps:  rows [0, lin0) = lin scalars+gates, padded to a tile,
// then 64 alpha rows.
template <int L> DEDF_HD constexpr int gates_total() { int g = 0; for (int l = 1; l <= L; ++l) g += mul_of(l); return g; }
template <int L> DEDF_HD constexpr int lin0_rows() { return mul_of(0) + gates_total<L>(); }          // 112 / 96
template <int L> DEDF_HD constexpr int alpha_row0() { return rup(lin0_rows<L>(), 32); }                // 128 / 96
template <int L> DEDF_HD constexpr int r0_tiles() { return (alpha_row0<L>() + mul_of(0)) / 32; }       // 6 / 5
DEDF_HD constexpr int gate_row(int l, int c) { int r = mul_of(0); for (int i = 1; i < l; ++i) r += mul_of(i); return r + c; }
// per-edge record written by the edge kernel: value in internal layout + one logit per head
template <int L> DEDF_HD constexpr int edge_rec() { return feat_dim<L>() + kHeads; }                   // 244 / 164

// ---- FFN row spaces (node kernel) ------------------------------------------------------------------------------------
// ---- layout of the node kernel's weight image -------------------------------------------------------------------------------
// Every block size depends on L only, so the offsets (in floats) are compile-time constants shared by the packer
// (dedf_pack.h::pack_node, which checks them) and the kernel: no offset ever occupies a scalar register.  (With run-time
// offsets the fully unrolled score stage kept > 300 scalars alive; hipcc spilled them to VGPR lanes / scratch.)
template <int L> struct NodeLayout {
    int A_proj[4], A_proj_l[4], ln_w[4], A_f1[4], A_f1_l[4], A_f2[4], A_f2_l[4];
    int b_proj0, ln_b0, b_f1, b_f2;
    int A_s[2][16], A_s_l[2][16], A_sl[2][2], A_sl_l[2][2], b_sl[2];
    int total;
};
template <int L> DEDF_HD constexpr int f1_rows0_fwd();
template <int L> DEDF_HD constexpr NodeLayout<L> make_node_layout() {
    NodeLayout<L> n{};
    int o = 0;
    auto img = [&](int O, int K) { const int sz = cdiv(O, 32) * cdiv(K, 16) * 256; const int at = o; o += sz; return at; };   // one hi or lo image
    auto rows = [&](int O) { const int at = o; o += cdiv(O, 32) * 32; return at; };
    for (int l = 0; l <= L; ++l) {
        const int m = mul_of(l), O1 = (l == 0 ? f1_rows0_fwd<L>() : hid_of(l)), Kh = hid_of(l);
        n.A_proj[l] = img(m, m); n.A_proj_l[l] = img(m, m);
        n.ln_w[l] = rows(m);
        n.A_f1[l] = img(O1, m); n.A_f1_l[l] = img(O1, m);
        n.A_f2[l] = img(m, Kh); n.A_f2_l[l] = img(m, Kh);
    }
    n.b_proj0 = rows(mul_of(0)); n.ln_b0 = rows(mul_of(0)); n.b_f1 = rows(f1_rows0_fwd<L>()); n.b_f2 = rows(mul_of(0));
    for (int tp = 0; tp < 2; ++tp) {
        for (int q = 0; q < stp_num_paths<L>(); ++q) {
            const PathInfo pi = stp_path<L>(q);
            n.A_s[tp][q] = img(pi.mul1, pi.mul2); n.A_s_l[tp][q] = img(pi.mul1, pi.mul2);
        }
        for (int l3 = 0; l3 < 2; ++l3) { n.A_sl[tp][l3] = img(mul_of(1), stp_k<L>(l3)); n.A_sl_l[tp][l3] = img(mul_of(1), stp_k<L>(l3)); }
        n.b_sl[tp] = rows(mul_of(1));
    }
    n.total = o;
    return n;
}

// accumulator -> true value of the node kernel's split-fp16 GEMMs (each weight matrix carries its own power-of-two scale,
// the B operands a fixed 2^kNodeBShift; dedf_pack.h::pack_node)
struct NodeScales {
    float proj[4], f1[4], f2[4], s[2][16], sl[2][2];
    // B-operand multipliers (powers of two) per stage, chosen per handle from the weights (dedf_pack.h::act_exponent): aggregate z -> proj,
    // normalised features -> fctp_1, hidden features of degree l -> fctp_2, field -> score TPs, contracted TP outputs -> final LinearRS
    float bz, bn, bh[4], bf, bt[2];
};
// floats per pose record: raw q [0:4], D^1 [4:13], D^2 [16:41], D^3 [48:97]
template <int L> DEDF_HD constexpr int pose_rec() { return L >= 3 ? 112 : 64; }
constexpr int kNodeBShift = 5;      // activation-side operand: typical magnitude 2^5 (dedf_pack.h::kActHeadroomBits)
constexpr int kMlpMid = 3;
template <int L> DEDF_HD constexpr int f1_rows0() { int r = hid_of(0); for (int l = 1; l <= L; ++l) r += hid_of(l); return r; }  // 288 / 336 / 368
template <int L> DEDF_HD constexpr int f1_rows0_fwd() { return f1_rows0<L>(); }
template <int L> inline constexpr NodeLayout<L> kNodeLayout = make_node_layout<L>();
DEDF_HD constexpr int f1_gate_row(int l, int c) { int r = hid_of(0); for (int i = 1; i < l; ++i) r += hid_of(i); return r + c; }

// internal feature layout: block l at blk_off(l), stored [m][mul]  (reference stores [mul][m])
DEDF_HD constexpr int int_idx(int l, int c, int m) { return blk_off(l) + m * mul_of(l) + c; }
DEDF_HD constexpr int ref_idx(int l, int c, int m) { return blk_off(l) + c * (2 * l + 1) + m; }

}  // namespace dedf
