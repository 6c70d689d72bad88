// The two fused kernels: persistent waves striding over tiles (tile count lives on the device: no host round trip).
// Their instantiations are compiled in separate translation units (dedf_kernels.hip, one -DDEDF_KUNIT=n each, built in
// parallel by __graft_entry__.build()); dedf_api.hip only declares them (extern template) unless DEDF_SINGLE_TU is defined.
#pragma once
#include <hip/hip_runtime.h>
#include "dedf_edge.h"
#include "dedf_node.h"

using namespace dedf;

template <int L, int F0, bool HP = false, int H1 = 128, int H2 = 64, bool UN = false, int MODE = 0, bool NW = false, bool SO2 = false, bool QT = false> __global__ __launch_bounds__(64, 1) void k_edge(EdgeParams P) {
    if (edge_gate_closed(P)) return;
    const int* ti = P.tile_info;
    const int ntiles = ti[P.n_scales];
    const Wave wv = make_wave(P.W, P.W_bytes);
#if defined(DEDF_PHASE_PROF)
    unsigned long long pacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr bool FRONT = front_rows_in_lds<L, MODE>();
    edge_rows_to_lds<L, H1, H2, FRONT>(P, wv);
    int enc_scale = -1;
    GeoPre geo{};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int scale = 0;
        while (t >= ti[scale + 1]) ++scale;
        if (scale != enc_scale) { edge_enc_to_lds<L, FRONT>(P, wv, scale); enc_scale = scale; }
        const int k = t - ti[scale];
        const int ebase = ti[16 + scale], En = ti[16 + scale + 1] - ebase;
        int e_next = -1;               // MODE 1: this lane's edge of the wave's next tile (padding lanes read the tile's first edge)
        if constexpr (MODE == 1) {
            const int tn = t + gridDim.x;
            if (tn < ntiles) {
                int sn = scale;
                while (tn >= ti[sn + 1]) ++sn;
                const int kn = tn - ti[sn], eb = ti[16 + sn], nv = min(32, ti[16 + sn + 1] - eb - 32 * kn);
                e_next = eb + 32 * kn + (wv.col < nv ? wv.col : 0);
            }
        }
#if defined(DEDF_PHASE_PROF)
        edge_tile<L, F0, HP, H1, H2, UN, MODE, NW, SO2, QT>(P, wv, scale, ebase + 32 * k, min(32, En - 32 * k), geo, e_next, pacc);
#else
        edge_tile<L, F0, HP, H1, H2, UN, MODE, NW, SO2, QT>(P, wv, scale, ebase + 32 * k, min(32, En - 32 * k), geo, e_next);
#endif
    }
#if defined(DEDF_PHASE_PROF)
    if (P.phase_prof && wv.lane == 0) for (int i = 0; i < 16; ++i) P.phase_prof[blockIdx.x * 16 + i] += pacc[i];
#endif
}
// Radial table of the sampler (EdgeParams::rtab): the front of the radial network on the length grid of every scale, computed by the edge
// tile's own code (edge_tile MODE 2), 32 grid nodes per tile.
template <int L, int F0, bool HP = false, int H1 = 128, int H2 = 64> __global__ __launch_bounds__(64, 1) void k_radial_table(EdgeParams P) {
    if (edge_gate_closed(P)) return;
    const Wave wv = make_wave(P.W, P.W_bytes);
    constexpr bool FRONT = front_rows_in_lds<L, 2>();      // (the LDS array edge_tile<..., MODE 2> reads)
    edge_rows_to_lds<L, H1, H2, FRONT>(P, wv);
    int enc_scale = -1;
    for (int t = blockIdx.x;; t += gridDim.x) {
        int scale = 0, base = 0;
        for (; scale < P.n_scales; ++scale) {
            const int nt = (P.rtab_n[scale] + 3 + 31) / 32;
            if (t - base < nt) break;
            base += nt;
        }
        if (scale >= P.n_scales) break;
        if (scale != enc_scale) { edge_enc_to_lds<L, FRONT>(P, wv, scale); enc_scale = scale; }
        const int k = t - base, nrows = P.rtab_n[scale] + 3;
#if defined(DEDF_PHASE_PROF)
        unsigned long long pacc[16];
        GeoPre geo{};
        edge_tile<L, F0, HP, H1, H2, false, 2>(P, wv, scale, 32 * k, min(32, nrows - 32 * k), geo, -1, pacc);
#else
        GeoPre geo{};
        edge_tile<L, F0, HP, H1, H2, false, 2>(P, wv, scale, 32 * k, min(32, nrows - 32 * k), geo, -1);
#endif
    }
}
// Accuracy check of the radial table (edge_tile MODE 3): 32 interval midpoints per tile, every interval of every scale.
template <int L, int F0, bool HP = false, int H1 = 128, int H2 = 64> __global__ __launch_bounds__(64, 1) void k_radial_check(EdgeParams P) {
    if (edge_gate_closed(P)) return;
    const Wave wv = make_wave(P.W, P.W_bytes);
    constexpr bool FRONT = front_rows_in_lds<L, 3>();
    edge_rows_to_lds<L, H1, H2, FRONT>(P, wv);
    int enc_scale = -1;
    for (int t = blockIdx.x;; t += gridDim.x) {
        int scale = 0, base = 0;
        for (; scale < P.n_scales; ++scale) {
            const int nt = (P.rtab_n[scale] + 31) / 32;
            if (t - base < nt) break;
            base += nt;
        }
        if (scale >= P.n_scales) break;
        if (scale != enc_scale) { edge_enc_to_lds<L, FRONT>(P, wv, scale); enc_scale = scale; }
        const int k = t - base;
        GeoPre geo{};
#if defined(DEDF_PHASE_PROF)
        unsigned long long pacc[16];
        edge_tile<L, F0, HP, H1, H2, false, 3>(P, wv, scale, 32 * k, min(32, P.rtab_n[scale] - 32 * k), geo, -1, pacc);
#else
        edge_tile<L, F0, HP, H1, H2, false, 3>(P, wv, scale, 32 * k, min(32, P.rtab_n[scale] - 32 * k), geo, -1);
#endif
    }
}
template <int L, bool EBM, bool HP = false, bool UN = false> __global__ __launch_bounds__(64, 1) void k_node(NodeParams P) {
    const Wave wv = make_wave(P.W, P.W_bytes);
    const int ntiles = (P.n_nodes + 31) / 32;
    node_rows_to_lds<L, EBM || UN>(P, wv);
    // (Measured and not kept, round 3: the per-pose reduction + Langevin update as a tail of this kernel -- the wave that completes a pose's
    //  last node reduces it, found with one atomic counter per pose.  The device-scope release / acquire that the hand-over needs is an L2
    //  write-back on this multi-XCD part: k_node 0.29 -> 0.45 ms on C2, 56 -> 84 us at 16 poses, against one 5 us launch saved;
    //  profiles/r03g_fused_reduce_and_single_block_fill_small_batch.log.)
    if constexpr (!EBM && !UN) {
        if (P.split) {      // two waves per tile, one score tensor product each (dedf_node.h: NodeParams::split)
            for (int t = blockIdx.x; t < 2 * ntiles; t += gridDim.x) node_tile<L, EBM, HP, UN>(P, wv, (t >> 1) * 32, t & 1);
            return;
        }
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) node_tile<L, EBM, HP, UN>(P, wv, t * 32);
}

#include "dedf_kernel_list.h"
