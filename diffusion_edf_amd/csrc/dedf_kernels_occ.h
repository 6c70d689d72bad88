// Occupancy variant of the fused edge kernel: the SAME tile code (dedf_edge.h::edge_tile) compiled for two waves per SIMD
// (__launch_bounds__(64, 2): at most 256 registers per wave).  At lmax 1 the tile's live state is small enough for that budget (the lmax-2
// tile needs 512 registers and 38 KB of LDS per wave); the LDS (22 KB of parked operands + 5 KB of row vectors per wave) then allows five
// waves per CU instead of four.  Kept as its own translation unit (dedf_kernels_occ.hip) so that the A/B against the one-wave kernel is one
// environment variable: DEDF_EDGE_OCC=1.
#pragma once
#include "dedf_kernels.h"

template <int L, int F0, bool HP = false, int H1 = 128, int H2 = 64> __global__ __launch_bounds__(64, 2) void k_edge_occ(EdgeParams P) {
    const int* ti = P.tile_info;
    const int ntiles = ti[P.n_scales];
    const Wave wv = make_wave(P.W, P.W_bytes);
    edge_rows_to_lds<L, H1, H2, true>(P, wv);
    int enc_scale = -1;
    GeoPre geo{};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int scale = 0;
        while (t >= ti[scale + 1]) ++scale;
        if (scale != enc_scale) { edge_enc_to_lds<L, true>(P, wv, scale); enc_scale = scale; }
        const int k = t - ti[scale];
        const int ebase = ti[16 + scale], En = ti[16 + scale + 1] - ebase;
#if defined(DEDF_PHASE_PROF)
        unsigned long long pacc[16] = {};
        edge_tile<L, F0, HP, H1, H2, false, 0>(P, wv, scale, ebase + 32 * k, min(32, En - 32 * k), geo, -1, pacc);
#else
        edge_tile<L, F0, HP, H1, H2, false, 0>(P, wv, scale, ebase + 32 * k, min(32, En - 32 * k), geo, -1);
#endif
    }
}
