// Per-pose reduction of the node kernel's outputs and the float64 Langevin step on SE(3) -- shared by the stand-alone reduction kernels
// (dedf_misc.h) and by the node kernel's fused tail (dedf_node.h: the wave that completes a pose's last query node carries its update).
#pragma once
#include "dedf_dev.h"

namespace dedf {

constexpr int kFlagOverflow = 41, kFlagNonFinite = 42;      // words of tile_info: sticky within ONE API call (cleared at its start)
// ... and the same two verdicts, sticky until the HOST has consumed them (dedf_api.hip::check_pending): a dedf_score / dedf_energy call never
// synchronises, so in a loop of back-to-back calls the per-call words of an earlier call are cleared before anyone has looked at them
constexpr int kFlagOverflowSeen = 56, kFlagNonFiniteSeen = 57;

// Sum over the query points of one pose (score_head.py:207-209), fixed order.
// One wave per pose: lanes stride over the query points, then a fixed butterfly (deterministic, independent of nT).
// Status words (tile_info): an edge-workspace overflow anywhere in this API call turns the outputs into NaN (the evaluation that
// overflowed produced nothing: stale results must never look valid); a non-finite sum (fp16 operand range exceeded, or NaN inputs)
// raises the sticky non-finite flag.
__device__ inline bool reduce_status(int* flags, float (&s)[6]) {
    const bool ovf = flags[kFlagOverflow] != 0;
    if (ovf) for (int i = 0; i < 6; ++i) s[i] = __builtin_nanf("");
    bool fin = true;
    for (int i = 0; i < 6; ++i) fin = fin && (fabsf(s[i]) <= 3.0e38f);
    if (!fin && !ovf) { flags[kFlagNonFinite] = 1; flags[kFlagNonFiniteSeen] = 1; }
    return ovf;
}
// ------------------------------------------------------------------------------------------------------------------------
// Langevin step on SE(3) in float64 (score_model_base.py:178-193).  Noise: caller-provided standard normals or
// Philox4x32-10 keyed by (seed, global pose index, step) + Box-Muller, so results do not depend on how poses are sharded.
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ inline double u01(uint32_t hi, uint32_t lo) {      // (0,1]
    const uint64_t v = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);   // 53 bits
    return ((double)(v & ((1ull << 53) - 1)) + 1.0) * (1.0 / 9007199254740992.0);
}
struct LangevinParams {
    double* T;                 // [nT][7] state, updated in place
    double t, alpha_ang, alpha_lin, temperature, ang_mult, lin_mult;
    const double* noise;       // [2][nT][3] for this step or nullptr
    uint64_t seed; int64_t first_pose; int step;
    double* traj_out;          // [nT][7] slot of this step
    int nT;
};
// update of pose i from its fp32 scores (ang, lin)
__device__ inline void langevin_update(const LangevinParams& P, int i, const float (&ang)[3], const float (&lin)[3]) {
    double* T = P.T + 7 * (size_t)i;
    double na[3], nl[3];
    if (P.noise) {
        for (int k = 0; k < 3; ++k) { na[k] = P.noise[(size_t)i * 3 + k]; nl[k] = P.noise[(size_t)(P.nT + i) * 3 + k]; }
    } else {
        const uint64_t gp = (uint64_t)(P.first_pose + i);
        double g[8];
        for (int b = 0; b < 2; ++b) {
            uint32_t r[8];
            philox4x32_10((uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)P.step, (uint32_t)(2 * b), (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r);
            philox4x32_10((uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)P.step, (uint32_t)(2 * b + 1), (uint32_t)P.seed, (uint32_t)(P.seed >> 32), r + 4);
            for (int p = 0; p < 2; ++p) {              // Box-Muller on two uniform pairs
                const double u1 = u01(r[4 * p], r[4 * p + 1]), u2 = u01(r[4 * p + 2], r[4 * p + 3]);
                const double rad = sqrt(-2.0 * log(u1)), th = 6.283185307179586476925 * u2;
                g[4 * b + 2 * p] = rad * cos(th); g[4 * b + 2 * p + 1] = rad * sin(th);
            }
        }
        for (int k = 0; k < 3; ++k) { na[k] = g[k]; nl[k] = g[3 + k]; }
    }
    const double st = sqrt(P.t);
    double da[3], dl[3];
    for (int k = 0; k < 3; ++k) {
        const double sa = (double)ang[k] / (P.ang_mult * st);
        const double sl = (double)lin[k] / (P.lin_mult * st);
        da[k] = (P.alpha_ang / 2) * sa + sqrt(P.temperature * P.alpha_ang) * na[k];
        dl[k] = (P.alpha_lin / 2) * sl + sqrt(P.temperature * P.alpha_lin) * nl[k];
    }
    const double q0 = T[0], q1 = T[1], q2 = T[2], q3 = T[3];
    // dq = L da,  L = T[q_indices] * q_factor  (score_model_base.py:31-32, 188-190)
    double dq[4];
    dq[0] = -0.5 * q1 * da[0] - 0.5 * q2 * da[1] - 0.5 * q3 * da[2];
    dq[1] = 0.5 * q0 * da[0] - 0.5 * q3 * da[1] + 0.5 * q2 * da[2];
    dq[2] = 0.5 * q3 * da[0] + 0.5 * q0 * da[1] - 0.5 * q1 * da[2];
    dq[3] = -0.5 * q2 * da[0] + 0.5 * q1 * da[1] + 0.5 * q0 * da[2];
    // dx = quaternion_apply(q, dl) with the pre-update q
    const double ow = -q1 * dl[0] - q2 * dl[1] - q3 * dl[2];
    const double ox = q0 * dl[0] + q2 * dl[2] - q3 * dl[1];
    const double oy = q0 * dl[1] - q1 * dl[2] + q3 * dl[0];
    const double oz = q0 * dl[2] + q1 * dl[1] - q2 * dl[0];
    const double rx = -ow * q1 + ox * q0 - oy * q3 + oz * q2;
    const double ry = -ow * q2 + ox * q3 + oy * q0 - oz * q1;
    const double rz = -ow * q3 - ox * q2 + oy * q1 + oz * q0;
    double n0 = q0 + dq[0], n1 = q1 + dq[1], n2 = q2 + dq[2], n3 = q3 + dq[3];
    const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    T[0] = n0 / nn; T[1] = n1 / nn; T[2] = n2 / nn; T[3] = n3 / nn;
    T[4] += rx; T[5] += ry; T[6] += rz;
    double* o = P.traj_out + 7 * (size_t)i;
    for (int k = 0; k < 7; ++k) o[k] = T[k];
}
// Sum over the query points of pose t in k_pose_reduce's order (lanes stride over the query points, then a fixed butterfly: deterministic,
// independent of nT) and the Langevin update of that pose; one WAVE, lane 0 carries the float64 update.
// (node_spin: the ang_spin half when the node kernel ran split over two waves per tile -- NodeParams::split --, else null)
__device__ inline void reduce_pose_and_step(const float* __restrict__ node_out, int nQ, int t, int lane, float* __restrict__ ang_out,
                                            float* __restrict__ lin_out, const LangevinParams& P, int* __restrict__ flags,
                                            const float* __restrict__ node_spin = nullptr) {
    float s[6] = {0, 0, 0, 0, 0, 0};
    for (int q = lane; q < nQ; q += 64) {
        const float* o = node_out + ((size_t)t * nQ + q) * 8;
        const f32x4 a = ld4(o), b = ld4(o + 4);
        if (node_spin != nullptr) {
            const f32x4 sp = ld4(node_spin + ((size_t)t * nQ + q) * 4);
            s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3] + sp[0]; s[4] += b[0] + sp[1]; s[5] += b[1] + sp[2];
        } else {
        s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3]; s[4] += b[0]; s[5] += b[1];
        }
    }
    for (int m = 32; m >= 1; m >>= 1)
        for (int i = 0; i < 6; ++i) s[i] += __shfl_xor(s[i], m, 64);
    if (lane != 0) return;
    reduce_status(flags, s);
    const float lin[3] = {s[0], s[1], s[2]}, ang[3] = {s[3], s[4], s[5]};
    for (int k = 0; k < 3; ++k) { lin_out[3 * t + k] = lin[k]; ang_out[3 * t + k] = ang[k]; }
    langevin_update(P, t, ang, lin);
}

}  // namespace dedf
