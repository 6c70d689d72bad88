// Compile-time description of the 16-EDGE tile of the fused edge kernel (dedf_edge16.h), lmax 2, shared by its host packers
// (dedf_pack16.h) and the device code.
//
// Why a second tiling: the 32-edge tile (dedf_edge.h) needs 512 registers and 38 KB of LDS per wave, i.e. ONE wave per SIMD, and is bound by
// VALU issue with nothing to hide its waits behind.  With v_mfma_f32_16x16x32_f16 a wave owns 16 edges: a lane is
// (edge column e = lane & 15, row group g = lane >> 4) and holds a QUARTER of an edge's channel rows, so every per-edge array is half as
// long per lane -- about 256 registers and 20-25 KB of LDS per wave: two waves per SIMD.
//
// Operand layouts of v_mfma_f32_16x16x32_f16 (checked by tests/probe/mfma16x32_probe.hip):
//   A [16 rows][32 k]:  lane holds A[row = lane & 15][k = 8 g + j], j = 0..7        (8 halves = 16 bytes)
//   B [32 k][16 cols]:  lane holds B[k = 8 g + j][col = e]
//   D [16 rows][16 cols]: lane holds D[row = 4 g + r][col = e], r = 0..3            (4 registers = one "row tile")
// Chaining: two consecutive 16-row output tiles (T, T + 1) of a layer are one K = 32 chunk of the next: the lane's 8 registers supply
// k = 8 g + j  <->  row 16 (T + (j >> 2)) + 4 g + (j & 3); the host permutes the weight columns accordingly (no data movement).
#pragma once
#include "dedf_net.h"

namespace dedf {

// row of a chunk's two tiles that element j of lane group g supplies: tile (j >> 2), row 4 g + (j & 3) inside it
DEDF_HD constexpr int chain16_tile(int j) { return j >> 2; }
DEDF_HD constexpr int chain16_row(int g, int j) { return 4 * g + (j & 3); }

// ---- walk of the first depth-wise TP: 16-channel weight tiles grouped by output degree l3, inside a group by creation order ------------
// (every path has 64 / 32 / 16 channels: a 16-row tile of the last radial layer belongs to exactly one path)
template <int L> struct Walk16 {
    int n;                       // tiles (= WN / 16)
    PathInfo path[64];           // path of tile t
    int u0[64];                  // first channel of the tile inside its path
    int grp0[5];                 // first tile of group l3 (grp0[L + 1] = n)
};
template <int L> DEDF_HD constexpr Walk16<L> make_walk16() {
    Walk16<L> w{};
    int n = 0;
    for (int l3 = 0; l3 <= L; ++l3) {
        w.grp0[l3] = n;
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            const PathInfo pi = dtp_path<L>(p);
            if (pi.l3 != l3) continue;
            for (int c = 0; c < pi.mul1 / 16; ++c) { w.path[n] = pi; w.u0[n] = 16 * c; ++n; }
        }
    }
    for (int l3 = L + 1; l3 < 5; ++l3) w.grp0[l3] = n;
    w.n = n;
    return w;
}
template <int L> inline constexpr Walk16<L> kWalk16 = make_walk16<L>();
template <int L> DEDF_HD constexpr int w16_tiles() { return kWalk16<L>.n; }
template <int L> DEDF_HD constexpr int w16_group_of(int t) { int l3 = 0; while (t >= kWalk16<L>.grp0[l3 + 1]) ++l3; return l3; }
template <int L> DEDF_HD constexpr int w16_group_tiles(int l3) { return kWalk16<L>.grp0[l3 + 1] - kWalk16<L>.grp0[l3]; }
template <int L> DEDF_HD constexpr int w16_group_chunks(int l3) { return cdiv(w16_group_tiles<L>(l3), 2); }
// e3nn weight row (row of the last radial layer / flat DTP weight) of walk row r
template <int L> DEDF_HD constexpr int w16_weight_row(int r) { return kWalk16<L>.path[r / 16].wstart + kWalk16<L>.u0[r / 16] + r % 16; }
// sorted DTP channel (inside the l3 block) of row `row` of walk tile t
template <int L> DEDF_HD constexpr int w16_channel(int t, int row) { return kWalk16<L>.path[t].kofs + kWalk16<L>.u0[t] + row; }

// row space of the l3 = 0 outputs of sep_act.lin + sep_alpha: [scalars | gates of l = 1.. | alpha], NO padding: 176 rows = 11 tiles (lmax 2)
template <int L> DEDF_HD constexpr int lin0_tiles16() { return (lin0_rows<L>() + mul_of(0)) / 16; }
template <int L> DEDF_HD constexpr int lin_tiles16(int l3) { return l3 == 0 ? lin0_tiles16<L>() : mul_of(l3) / 16; }
// first slot of chunk q of group l3 in the lin A stream (slot = one (chunk, output tile) operand: hi image | lo image, 512 floats)
template <int L> DEDF_HD constexpr int lin16_slot(int l3, int q) {
    int s = 0;
    for (int a = 0; a < l3; ++a) s += w16_group_chunks<L>(a) * lin_tiles16<L>(a);
    return s + q * lin_tiles16<L>(l3);
}
template <int L> DEDF_HD constexpr int lin16_slots() { return lin16_slot<L>(L + 1, 0); }

// ---- value GEMMs (second depth-wise TP in output-side form, see dedf_net.h::make_val_walk) -----------------------------------------------
// gated features as K = 32 chunks: degree 0: 64 channels = 2 chunks; degree 1: 32 = 1 chunk per component; degree 2: 16 = half a chunk
DEDF_HD constexpr int feat_chunks16(int l) { return cdiv(mul_of(l), 32); }
// parked slot (B operand chunk) of (degree l, component i, chunk c)
template <int L> DEDF_HD constexpr int park16_slot(int l, int i, int c) {
    int q = 0;
    for (int a = 0; a < l; ++a) q += (2 * a + 1) * feat_chunks16(a);
    return q + i * feat_chunks16(l) + c;
}
template <int L> DEDF_HD constexpr int park16_slots() { return park16_slot<L>(L + 1, 0, 0); }      // 10 at lmax 2
DEDF_HD constexpr int val_tiles16(int l3) { return mul_of(l3) / 16; }
// first A slot of (path p, input chunk c, output tile To): paths in creation order grouped by l3
template <int L> DEDF_HD constexpr int val16_slot(int p, int c, int To) {
    int s = 0;
    for (int l3 = 0; l3 <= L; ++l3)
        for (int q = 0; q < dtp_num_paths<L>(); ++q) {
            const PathInfo pi = dtp_path<L>(q);
            if (pi.l3 != l3) continue;
            if (q == p) return s + c * val_tiles16(l3) + To;
            s += feat_chunks16(pi.l1) * val_tiles16(l3);
        }
    return s;
}
template <int L> DEDF_HD constexpr int val16_slots() { return val16_slot<L>(-1, 0, 0); }
// work items of the value stage in the order they run: (path, output tile), paths grouped by output degree
struct VItem16 { int p, To; };
template <int L> DEDF_HD constexpr int val16_num_items() {
    int n = 0;
    for (int p = 0; p < dtp_num_paths<L>(); ++p) n += val_tiles16(dtp_path<L>(p).l3);
    return n;
}
template <int L> DEDF_HD constexpr VItem16 val16_item(int I) {
    int n = 0;
    for (int l3 = 0; l3 <= L; ++l3)
        for (int p = 0; p < dtp_num_paths<L>(); ++p) {
            if (dtp_path<L>(p).l3 != l3) continue;
            for (int To = 0; To < val_tiles16(l3); ++To, ++n) if (n == I) return VItem16{p, To};
        }
    return VItem16{-1, 0};
}
// first / last item of output degree l3
template <int L> DEDF_HD constexpr int val16_group_first(int l3) {
    for (int I = 0; I < val16_num_items<L>(); ++I) if (dtp_path<L>(val16_item<L>(I).p).l3 == l3) return I;
    return val16_num_items<L>();
}
// channel of the path's input block that element (g, j) of chunk c supplies, or -1 (padding: the upper half of a 16-channel block's chunk)
DEDF_HD constexpr int feat16_channel(int l, int c, int g, int j) {
    const int ch = 16 * (2 * c + chain16_tile(j)) + chain16_row(g, j);
    return ch < mul_of(l) ? ch : -1;
}

}  // namespace dedf
