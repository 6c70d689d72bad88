// Register-fragment layout shared by host packers and device kernels.
//
// All dense per-edge / per-node layers are computed "transposed":   D[out][item] = sum_k W[out][k] * act[k][item]
// with v_mfma_f32_32x32x2_f32: A = weights (32 out rows x 2 k), B = activations (2 k x 32 items), so that one
// lane owns ONE item (edge or query node, column = lane & 31) and half of the channel rows (hi = lane >> 5).
// A 32x32 output tile lives in 16 accumulator registers per lane; register r of a lane with half `hi`
// holds row  rowmap(r, hi) = (r & 3) + 8 * (r >> 2) + 4 * hi   (gfx950 C/D map, MI355X guide §3).
// Feeding a layer's output into the next layer needs NO data movement: K-step (tile T, reg r) of the next
// MFMA takes B = acc[T][r] directly — lanes hi=0 supply k = 32T+rowmap(r,0), lanes hi=1 supply 32T+rowmap(r,1)
// — and the host pre-permutes the weight matrix so that lane l loads A = W[out0 + (l & 31)][k(l >> 5)].
#pragma once
#include <cstdint>
#include <vector>

#if defined(__HIPCC__) || defined(__HIP__)
#define DEDF_HD __host__ __device__ __forceinline__
#else
#define DEDF_HD inline
#endif

namespace dedf {

DEDF_HD constexpr int rowmap(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// inverse: which (reg, hi) holds row `row` (0..31) of a tile
DEDF_HD constexpr int row_reg(int row) { return (row & 3) + 4 * (row >> 3); }
DEDF_HD constexpr int row_hi(int row) { return (row >> 2) & 1; }

// A K-step: the k index supplied by the lower / upper half-wave (or -1 = zero padding).
struct KStep { int k0, k1; };

// Pack W (O x K, row-major with leading dimension ld, optionally transposed access via a functor) into the
// order lanes consume it: [out tile To][step group g = s/4][lane 0..63][j = s%4]  (one float4 per lane and
// group; 1 KiB per wave-instruction, fully coalesced).
template <class WAt>   // WAt(o, k) -> float
inline std::vector<float> pack_A(int O, const std::vector<KStep>& steps, WAt W) {
    const int nTo = (O + 31) / 32;
    const int nG = ((int)steps.size() + 3) / 4;
    std::vector<float> out((size_t)nTo * nG * 64 * 4, 0.0f);
    for (int To = 0; To < nTo; ++To)
        for (int s = 0; s < (int)steps.size(); ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = To * 32 + (lane & 31);
                const int k = (lane >> 5) ? steps[s].k1 : steps[s].k0;
                float v = 0.0f;
                if (o < O && k >= 0) v = W(o, k);
                out[(((size_t)To * nG + s / 4) * 64 + lane) * 4 + (s & 3)] = v;
            }
    return out;
}

// Split-fp16 A images for v_mfma_f32_32x32x16_f16 (3-term products hi*hi + hi*lo + lo*hi reproduce fp32 products to ~2^-22):
// a K-chunk = 16 k's = 8 accumulator registers of each half-wave; lane (i = l & 31, h = l >> 5) supplies
// A[out i][k = kof(chunk, j, h)], j = 0..7, as 8 halves (16 B).  Image: [out tile][chunk][lane 64][8 halves], one for the high
// halves and one for the residuals; returned as raw bits in float vectors (4 floats per lane per chunk).
template <class WAt, class KOf>
inline void pack_A_h(int O, int nch, WAt W, KOf kof, std::vector<float>& img_hi, std::vector<float>& img_lo) {
    const int nTo = (O + 31) / 32;
    std::vector<uint16_t> H((size_t)nTo * nch * 64 * 8, 0), Lo((size_t)nTo * nch * 64 * 8, 0);
    for (int To = 0; To < nTo; ++To)
        for (int c = 0; c < nch; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = To * 32 + (lane & 31), k = kof(c, j, lane >> 5);
                    if (o >= O || k < 0) continue;
                    const float w = W(o, k);
                    const _Float16 hh = (_Float16)w;
                    const _Float16 ll = (_Float16)(w - (float)hh);
                    const size_t idx = (((size_t)To * nch + c) * 64 + lane) * 8 + j;
                    __builtin_memcpy(&H[idx], &hh, 2);
                    __builtin_memcpy(&Lo[idx], &ll, 2);
                }
    img_hi.assign(H.size() / 2, 0.0f);
    img_lo.assign(H.size() / 2, 0.0f);
    __builtin_memcpy(img_hi.data(), H.data(), H.size() * 2);
    __builtin_memcpy(img_lo.data(), Lo.data(), Lo.size() * 2);
}
// k of element j of chunk c when the B operand is a producer's accumulator tiles (row layout), K valid rows
inline int chain_k(int K, int c, int j, int h) { const int k = 32 * (c / 2) + rowmap(8 * (c % 2) + j, h); return k < K ? k : -1; }

// K-steps that read a producer's accumulator tiles in order: K rows -> ceil(K/8) groups of 4 steps (a partial last
// tile only contributes the register groups that hold valid rows, exactly as the kernels walk them).
inline std::vector<KStep> chain_steps(int K) {
    std::vector<KStep> s;
    const int nT = (K + 31) / 32;
    for (int T = 0; T < nT; ++T)
        for (int g = 0; g < 4 && 32 * T + 8 * g < K; ++g)
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                int k0 = 32 * T + rowmap(r, 0), k1 = 32 * T + rowmap(r, 1);
                s.push_back({k0 < K ? k0 : -1, k1 < K ? k1 : -1});
            }
    return s;
}

// Per-row vectors (bias, LN affine) in the order a lane reads them: [tile][hi][r]  (16 contiguous floats).
template <class VAt>
inline std::vector<float> pack_rows(int O, VAt V) {
    const int nTo = (O + 31) / 32;
    std::vector<float> out((size_t)nTo * 32, 0.0f);
    for (int To = 0; To < nTo; ++To)
        for (int hi = 0; hi < 2; ++hi)
            for (int r = 0; r < 16; ++r) {
                const int o = To * 32 + rowmap(r, hi);
                if (o < O) out[((size_t)To * 2 + hi) * 16 + r] = V(o);
            }
    return out;
}

}  // namespace dedf
