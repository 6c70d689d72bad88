// The small kernels around the two fused ones: pose preparation (Wigner-D, transformed query positions), time
// embedding -> pre-linear bias rows, pose-independent source message, radius neighbour search (count / scan / fill),
// joint softmax + aggregation over all scales, per-pose reduction and the float64 Langevin update.
#pragma once
#include "dedf_dev.h"
#include "dedf_net.h"
#include "dedf_node.h"
#include "dedf_langevin.h"

namespace dedf {

// ------------------------------------------------------------------------------------------------------------------------
// Pose preparation: one block per pose.
//   pose record: raw q, D^1(q), D^2(q)   with  q_n = standardize(q/|q|), R = quaternion_to_matrix(q_n),
//   (a,b,c) = matrix_to_euler_angles(R, "YXY"), D^l = X(a) J X(b) J X(c)            (wigner.py:44-81, 257-283;
//   transforms.py:83-110, 198-208, 271-308 — including the signed-zero behaviour at q = identity)
//   qpos[pose][q] = quaternion_apply(q_raw, x_q) + t                                  (gnn_data.py:95)
template <int LD>
__device__ inline void wigner_from_angles(float a, float b, float c, float* D /* (2l+1)^2 row-major */) {
    constexpr int n = 2 * LD + 1;
    float X[3][n][n];
    const float ang[3] = {a, b, c};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < n; ++j) X[t][i][j] = 0.0f;
#pragma unroll
        for (int i = 0; i < n; ++i) {                      // wigner.py:21-42 (sin first, cos overwrites the centre)
            const float f = (float)(LD - i);
            X[t][i][n - 1 - i] = sinf(f * ang[t]);
            X[t][i][i] = cosf(f * ang[t]);
        }
    }
    float A[n][n], Bm[n][n];
    auto Jl = [](int i, int j) { if constexpr (LD == 1) return kJ1[i][j]; else if constexpr (LD == 2) return kJ2[i][j]; else return kJ3[i][j]; };
    // ((((Xa J) Xb) J) Xc)
    // every loop has constant bounds and is unrolled, so that the matrices live in registers (they went to scratch otherwise)
#define DEDF_MM(OUT, LHS, RHS)                                   \
    _Pragma("unroll") for (int i = 0; i < n; ++i)               \
    _Pragma("unroll") for (int j = 0; j < n; ++j) {             \
        float s = 0;                                            \
        _Pragma("unroll") for (int k = 0; k < n; ++k) s += LHS * RHS; \
        OUT = s;                                                \
    }
    DEDF_MM(A[i][j], X[0][i][k], Jl(k, j))
    DEDF_MM(Bm[i][j], A[i][k], X[1][k][j])
    DEDF_MM(A[i][j], Bm[i][k], Jl(k, j))
    DEDF_MM(D[i * n + j], A[i][k], X[2][k][j])
#undef DEDF_MM
}

// pose record of one pose: raw q, D^1(q), D^2(q) (, D^3(q))
template <int L>
__device__ inline void pose_record(const float (&T)[7], float* __restrict__ rec) {
    const float qw = T[0], qi = T[1], qj = T[2], qk = T[3];
    rec[0] = qw; rec[1] = qi; rec[2] = qj; rec[3] = qk;
    const float nrm = sqrtf(qw * qw + qi * qi + qj * qj + qk * qk);       // torch.norm
    float r = qw / nrm, i = qi / nrm, j = qj / nrm, k = qk / nrm;
    if (r < 0.0f) { r = -r; i = -i; j = -j; k = -k; }                      // standardize_quaternion
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    // rows of R (transforms.py:96-108); only row 1 / column 1 are needed for YXY
    const float R01 = two_s * (i * j - k * r);
    const float R10 = two_s * (i * j + k * r);
    const float R11 = 1.0f - two_s * (i * i + k * k);
    const float R12 = two_s * (j * k - i * r);
    const float R21 = two_s * (j * k + i * r);
    const float a = atan2f(R01, R21);              // _angle_from_tan("Y","X", R[:,1], horizontal=False)
    const float b = acosf(R11);
    const float c = atan2f(R10, -R12);             // _angle_from_tan("Y","X", R[1,:], horizontal=True)
    wigner_from_angles<1>(a, b, c, rec + 4);
    if constexpr (L >= 2) wigner_from_angles<2>(a, b, c, rec + 16);
    if constexpr (L >= 3) wigner_from_angles<3>(a, b, c, rec + 48);
}
// quaternion_apply(q, p) + t = (q * (0,p)) * conj(q) + t   with the reference's raw products (transforms.py:113-165)
__device__ inline void pose_apply(const float (&T)[7], float px, float py, float pz, float& ox_, float& oy_, float& oz_) {
    const float qw = T[0], qi = T[1], qj = T[2], qk = T[3];
    const float aw = qw, ax = qi, ay = qj, az = qk;
    const float bw = 0.0f, bx = px, by = py, bz = pz;
    const float ow = aw * bw - ax * bx - ay * by - az * bz;
    const float ox = aw * bx + ax * bw + ay * bz - az * by;
    const float oy = aw * by - ax * bz + ay * bw + az * bx;
    const float oz = aw * bz + ax * by - ay * bx + az * bw;
    const float cw = qw, cx = -qi, cy = -qj, cz = -qk;
    const float rx = ow * cx + ox * cw + oy * cz - oz * cy;
    const float ry = ow * cy - ox * cz + oy * cw + oz * cx;
    const float rz = ow * cz + ox * cy - oy * cx + oz * cw;
    ox_ = rx + T[4]; oy_ = ry + T[5]; oz_ = rz + T[6];
}
// Ts64 != nullptr (sampler): the float64 state is read directly and rounded to fp32 here, exactly what a separate cast would give.
__device__ inline void load_pose(const float* __restrict__ Ts, const double* __restrict__ Ts64, int t, float (&T)[7]) {
    for (int k = 0; k < 7; ++k) T[k] = Ts64 != nullptr ? (float)Ts64[7 * (size_t)t + k] : Ts[7 * (size_t)t + k];
}
template <int L>
__global__ void k_pose_prep(const float* __restrict__ Ts, const double* __restrict__ Ts64, const float* __restrict__ qx, int nQ,
                            float* __restrict__ pose, float* __restrict__ qpos) {
    const int t = blockIdx.x;
    float T[7];
    load_pose(Ts, Ts64, t, T);
    if (threadIdx.x == 0) pose_record<L>(T, pose + (size_t)t * pose_rec<L>());
    for (int q = threadIdx.x; q < nQ; q += blockDim.x) {
        float* o = qpos + ((size_t)t * nQ + q) * 3;
        pose_apply(T, qx[3 * q], qx[3 * q + 1], qx[3 * q + 2], o[0], o[1], o[2]);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Time embedding (score_head.py:159-164) folded into the pre-linear (multiscale_tensor_field.py:225-234):
//   tb[p][n][row-packed 128] = W_pre[n][:, 64:] . MLP_n(sinusoid(time_p)) + b_pre[n]
// grid (nTb, n_scales), block 128.
struct TimeParams {
    const float* time; int time_stride;             // stride 0: one shared time
    const float *w1, *b1, *w2, *b2;                 // [n_scales][H][E], [H], [TE][H], [TE]
    const float *wpre, *bpre;                       // [n_scales][F0][F0], [F0]   (F0 = 64 + TE)
    int E, H, TE;                                   // time_emb_mlp = [E, H, TE]: [256,128,64] or [512,256,128]
    float max_time, time_enc_n;
    float* tb;
    int* varies;          // optional: set to 1 when a row's time differs from row 0's (tile_info[kFlagTimeVaries]: dedf_score's launch gate)
};
constexpr int kFlagTimeVaries = 45;      // word of tile_info, cleared at the start of every API call
constexpr int kTimeMaxEnc = 512, kTimeMaxHid = 256, kTimeMaxEmb = 128;
// The time path runs in FLOAT64 (round 6).  The sinusoid's argument reaches time_enc_n = 10 000 rad, where a float32 has an ulp of 1e-3 rad: in
// float32 -- the reference's own arithmetic included -- the high-frequency channels carry an error of ~5e-4, which query_time_encoding passes
// straight into every edge message (the fp32 restatement's destination message sits 6.6e-5 from the fp64 one; with edge time encoding alone the
// pre-linear averages it down to 7e-6).  One row per pose / per step and ~50 k multiply-adds per row: float64 costs nothing here, and the rows are
// exact functions of the float32 `time` the caller hands over (the reference's `time` tensor is float32 too, score_model_base.py:177).
__device__ __forceinline__ void time_sinusoid(double t, double max_time, double n, int E, double* enc) {      // radial_func.py:291-316, exactly
    const double x = t / max_time * n, step = log(n) / (double)(E / 2 - 1);
    for (int i = threadIdx.x; i < E / 2; i += blockDim.x) {
        const double a = x * exp(-(double)i * step);
        enc[i] = sin(a);
        enc[i + E / 2] = cos(a);
    }
}
__global__ __launch_bounds__(256) void k_time_bias(TimeParams P) {
    __shared__ double enc[kTimeMaxEnc], hid[kTimeMaxHid], emb[kTimeMaxEmb];
    const int p = blockIdx.x, n = blockIdx.y, n_scales = gridDim.y, tid = threadIdx.x;
    const int E = P.E, H = P.H, TE = P.TE, F0 = kLenEmb + TE;
    const double t = (double)P.time[p * P.time_stride];
    if (P.varies != nullptr && n == 0 && tid == 0 && !(t == (double)P.time[0])) *P.varies = 1;       // (a NaN time reads as "varies")
    time_sinusoid(t, (double)P.max_time, (double)P.time_enc_n, E, enc);
    __syncthreads();
    for (int i = tid; i < H; i += blockDim.x) {
        const float* w = P.w1 + ((size_t)n * H + i) * E;
        double s = P.b1[n * H + i];
        for (int k = 0; k < E; ++k) s += (double)w[k] * enc[k];
        hid[i] = s / (1.0 + exp(-s));
    }
    __syncthreads();
    for (int i = tid; i < TE; i += blockDim.x) {
        const float* w = P.w2 + ((size_t)n * TE + i) * H;
        double s = P.b2[n * TE + i];
        for (int k = 0; k < H; ++k) s += (double)w[k] * hid[k];
        emb[i] = s;
    }
    __syncthreads();
    for (int i = tid; i < F0; i += blockDim.x) {
        const float* w = P.wpre + ((size_t)n * F0 + i) * F0 + kLenEmb;
        double s = P.bpre[n * F0 + i];
        for (int k = 0; k < TE; ++k) s += (double)w[k] * emb[k];
        const int tile = i >> 5, row = i & 31;
        P.tb[((size_t)p * n_scales + n) * F0 + (tile * 2 + row_hi(row)) * 16 + row_reg(row)] = (float)s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// query_time_encoding (score_head.py:64-70, 168-173): every query point of a pose carries  e = query_time_mlp(sinusoid(time))  -- TE scalars -- as the
// DESTINATION feature of the key field's block (use_dst_feature, gnn_block.py:109-130).  Everything the block does with it depends on the time alone,
// so it is evaluated once per time row p (a pose of dedf_score, a step of dedf_sample) instead of once per (pose, query point):
//   rows[p][0:64]   = out_scale * (linear_dst(prenorm_dst(e)) + bias)     joins the 0e block of every edge message of the pose (gnn_block.py:172-180;
//                     natural channel order, scaled like the stored source message: dedf_pack.h::EdgeOffsets::msg_scale)
//   rows[p][64:128] = skip_1(e) = LinearRS(e) + bias                      joins the 0e block of the attention output (gnn_block.py:111, 205-206;
//                     row-packed like the node kernel's bias rows)
// grid (rows), block 256.
constexpr int kQueryTimeRow = 128;      // floats per time row
struct TimeQueryParams {
    const float* time; int time_stride;
    const float *w1, *b1, *w2, *b2;                 // query_time_mlp: [H][E], [H], [TE][H], [TE]
    const float *ln_w, *ln_b;                       // prenorm_dst affine [TE], [TE]
    const float *wdst, *bdst, *wskip, *bskip;       // LinearRS 0e -> 0e: [TE][64] (in x out), [64]
    int E, H, TE;
    float max_time, time_enc_n, out_scale;
    float* rows;
};
__global__ __launch_bounds__(256) void k_time_query(TimeQueryParams P) {      // (float64 like k_time_bias: see there)
    __shared__ double enc[kTimeMaxEnc], hid[kTimeMaxHid], emb[kTimeMaxEmb], nrm[kTimeMaxEmb], red[2];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int E = P.E, H = P.H, TE = P.TE;
    const double t = (double)P.time[p * P.time_stride];
    time_sinusoid(t, (double)P.max_time, (double)P.time_enc_n, E, enc);
    __syncthreads();
    for (int i = tid; i < H; i += blockDim.x) {
        const float* w = P.w1 + (size_t)i * E;
        double s = P.b1[i];
        for (int k = 0; k < E; ++k) s += (double)w[k] * enc[k];
        hid[i] = s / (1.0 + exp(-s));
    }
    __syncthreads();
    for (int i = tid; i < TE; i += blockDim.x) {
        const float* w = P.w2 + (size_t)i * H;
        double s = P.b2[i];
        for (int k = 0; k < H; ++k) s += (double)w[k] * hid[k];
        emb[i] = s;
    }
    __syncthreads();
    if (tid == 0) {      // EquivariantLayerNormV2 on TE x 0e (layer_norm.py:113-146): mean-free, 'component' norm over the block, eps 1e-5
        double mean = 0.0;
        for (int k = 0; k < TE; ++k) mean += emb[k];
        mean /= TE;
        double v = 0.0;
        for (int k = 0; k < TE; ++k) { const double d = emb[k] - mean; v += d * d; }
        red[0] = mean; red[1] = 1.0 / sqrt(v / TE + 1e-5);
    }
    __syncthreads();
    for (int i = tid; i < TE; i += blockDim.x) nrm[i] = (emb[i] - red[0]) * (red[1] * (double)P.ln_w[i]) + (double)P.ln_b[i];
    __syncthreads();
    constexpr int M0 = mul_of(0);
    float* const o = P.rows + (size_t)p * kQueryTimeRow;
    for (int i = tid; i < 2 * M0; i += blockDim.x) {
        const int w = i % M0;
        if (i < M0) {
            double s = P.bdst[w];
            for (int u = 0; u < TE; ++u) s += (double)P.wdst[u * M0 + w] * nrm[u];
            o[w] = (float)(s * (double)P.out_scale);
        } else {
            double s = P.bskip[w];
            for (int u = 0; u < TE; ++u) s += (double)P.wskip[u * M0 + w] * emb[u];
            const int tile = w >> 5, row = w & 31;
            o[M0 + (tile * 2 + row_hi(row)) * 16 + row_reg(row)] = (float)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Source message (pose independent): EquivariantLayerNormV2 + LinearRS(bias) on every key point
// (gnn_block.py:170-171, layer_norm.py:91-156, tensor_product_rescale.py:176-185).  One block (64 threads) per point.
// NORM = false: LinearRS only, `bias` may be null (the linear_src / linear_dst of a UNet block: block.py:149-153 overwrites the
// LayerNorm outputs, so the linears see the raw features).
// v0..v3: true multiplicities for the LayerNorm statistics of zero-padded models (0 = all channels): the padded channels are exactly 0
// and their affine weights are 0; the mean is taken over the true channels and the mean^2 each padded 0e channel adds to the variance
// sum is taken out again.
// TRUE_IN (score head at lmax 3): features, LayerNorm affine and weights come in the reference's TRUE shapes (8x3e); the message leaves in
// the kernel layout (16x3e, true channels at dedf_net.h::pad_pos, the others 0).  Otherwise input and output are both in the kernel layout.
template <int L, bool NORM = true, bool TRUE_IN = false>
__global__ void k_src_message(const float* __restrict__ f, int n_pts, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                              const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ msg,
                              int v0 = 0, int v1 = 0, int v2 = 0, int v3 = 0, float out_scale = 1.0f) {
    // out_scale: power of two the message is stored with (score heads: EdgeOffsets::msg_scale -- the activation-side exponent of the edge kernel's
    // B operands  w x'  rides on the message, so that the last radial layer's weight image keeps its own, weight-side scale)
    constexpr int Din = TRUE_IN ? true_feat_dim<L>() : feat_dim<L>(), Dout = feat_dim<L>();
    auto m_in = [](int l) { return TRUE_IN ? true_mul(l) : mul_of(l); };
    auto off_in = [](int l) { return TRUE_IN ? true_blk_off(l) : blk_off(l); };
    __shared__ float x[Din], red[2];
    const int pt = blockIdx.x, tid = threadIdx.x;
    const float* fi = f + (size_t)pt * Din;
    for (int i = tid; i < Din; i += 64) x[i] = fi[i];
    __syncthreads();
    int woff = 0, chan = 0;
    if constexpr (NORM) for (int l = 0; l <= L; ++l) {
        const int m = m_in(l), d = 2 * l + 1, off = off_in(l);
        // statistics of block l (serial in thread 0: a few hundred values, runs once per scene)
        if (tid == 0) {
            const int vl = l == 0 ? v0 : (l == 1 ? v1 : (l == 2 ? v2 : v3)), mv = vl > 0 ? vl : m;
            float mean = 0.0f;
            if (l == 0) { for (int c = 0; c < m; ++c) mean += x[off + c]; mean /= mv; }
            float v = 0.0f;
            for (int c = 0; c < m; ++c) {
                float s = 0.0f;
                for (int k = 0; k < d; ++k) { const float t = x[off + c * d + k] - mean; s += t * t; }
                v += s / d;
            }
            v = fmaxf(v - (float)(m - mv) * (mean * mean), 0.0f);
            red[0] = mean; red[1] = 1.0f / sqrtf(v / mv + 1e-5f);
        }
        __syncthreads();
        const float mean = red[0], rs = red[1];
        __syncthreads();
        for (int i = tid; i < m * d; i += 64) {
            const int c = i / d;
            float t = (x[off + i] - mean) * (rs * ln_w[chan + c]);
            if (l == 0) t += ln_b[c];
            x[off + i] = t;
        }
        __syncthreads();
        woff += m * m; chan += m;
    }
    woff = 0;
    float* o = msg + (size_t)pt * Dout;
    for (int l = 0; l <= L; ++l) {
        const int m = m_in(l), d = 2 * l + 1, off = off_in(l), mo = mul_of(l);
        if (TRUE_IN && mo != m) for (int i = tid; i < mo * d; i += 64) o[blk_off(l) + i] = 0.0f;      // the padded channels
        if (TRUE_IN && mo != m) __syncthreads();
        for (int i = tid; i < m * d; i += 64) {
            const int w = i / d, k = i % d;
            float s = (l == 0 && bias != nullptr) ? bias[w] : 0.0f;
            for (int u = 0; u < m; ++u) s += W[woff + u * m + w] * x[off + u * d + k];
            o[blk_off(l) + (TRUE_IN ? pad_pos(l, w) : w) * d + k] = s * out_scale;
        }
        woff += m * m;
    }
}

// (N, true dim) features in the reference layout -> (N, kernel dim): true channel c of degree l at pad_pos(l, c), the others 0
template <int L>
__global__ void k_pad_features(const float* __restrict__ in, float* __restrict__ out, int n) {
    constexpr int D = feat_dim<L>(), Dt = true_feat_dim<L>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * D) return;
    const int node = (int)(i / D), k = (int)(i % D);
    int l = 0;
    while (l < L && k >= blk_off(l + 1)) ++l;
    const int d = 2 * l + 1, r = k - blk_off(l), c = r / d, m = r % d;
    float v = 0.0f;
    for (int ct = 0; ct < true_mul(l); ++ct) if (pad_pos(l, ct) == c) v = in[(size_t)node * Dt + true_blk_off(l) + ct * d + m];
    out[i] = v;
}

// ------------------------------------------------------------------------------------------------------------------------
// dedf_field: node features from the node kernel's internal layout [l][m][channel] (kernel multiplicities) to the reference layout
// [l][channel][m] with the TRUE multiplicities
template <int L>
__global__ void k_internal_to_ref(const float* __restrict__ in, float* __restrict__ out, int n) {
    constexpr int D = feat_dim<L>(), Dt = true_feat_dim<L>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * Dt) return;
    const int node = (int)(i / Dt), k = (int)(i % Dt);
    int l = 0;
    while (l < L && k >= true_blk_off(l + 1)) ++l;
    const int d = 2 * l + 1, r = k - true_blk_off(l), c = r / d, m = r % d;
    out[i] = in[(size_t)node * D + blk_off(l) + m * mul_of(l) + pad_pos(l, c)];
}

// dedf_keypoint_weight (keypoint_extractor.py:111-119,185-194): one wave per point, lane = scalar channel
__global__ void k_keypoint_weight(const float* __restrict__ field, const float* __restrict__ emb, int n, int stride, const float* __restrict__ skip_W,
                                  const float* __restrict__ skip_b, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                  const float* __restrict__ lin_w, float lin_b, int sigmoid, float mult, float* __restrict__ out) {
    const int pt = blockIdx.x * 4 + threadIdx.x / 64, c = threadIdx.x % 64;
    if (pt >= n) return;
    const float* e = emb + (size_t)pt * stride;
    float pre = field[(size_t)pt * stride + c] - e[c] + skip_b[c];
    for (int u = 0; u < 64; ++u) pre += skip_W[u * 64 + c] * e[u];
    auto wave_sum = [](float v) {
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const float mean = wave_sum(pre) * (1.0f / 64);
    const float dv = pre - mean;
    const float var = wave_sum(dv * dv) * (1.0f / 64);
    float y = dv / sqrtf(var + 1e-5f) * ln_w[c] + ln_b[c];
    y = y / (1.0f + expf(-y));
    float w = wave_sum(y * lin_w[c]) + lin_b;
    if (sigmoid) w = 1.0f / (1.0f + expf(-w));
    if (c == 0) out[pt] = w * mult;
}

// ------------------------------------------------------------------------------------------------------------------------
// Radius neighbour search (torch_cluster.radius as used at graph_parser.py:339; all pairs for the infinite scale,
// graph_parser.py:279-281).  Keys are tiny (<= a few thousand points) and static, queries move every step: brute force with
// the key cloud streamed through LDS, one thread per destination node.  Edge order: scale, then dst, then src ascending.
//   pass 1 (count):  cnt[n][d], neighbour bit masks and per-block totals;
//   pass 2 (fill):   block-local exclusive scan of cnt + block offset -> off[n][d], then the edge lists.
struct NbrParams {
    const float* key_x; int n_keys;
    int scale_start[kMaxScales + 1];
    float r2[kMaxScales];                 // squared radius, <= 0: infinite
    int n_scales, max_neighbors;
    const float* qpos; int n_dst;
    int* cnt;                             // [n_scales][n_dst]
    int* off;                             // [n_scales][n_dst]  exclusive prefix inside the scale
    int* blk;                             // [n_scales][n_blocks] block totals of the count pass
    int* tile_info;                       // written by block 0 of the fill pass: [0..n_scales] tile prefix, [16..16+n_scales] edge prefix, [40] overflow
                                          // of this evaluation, [41] sticky overflow (set here, cleared by the host at the start of an API call),
                                          // [42] sticky non-finite score (set by the per-pose reductions)
    long long* edge_hist;                 // optional running edge count (statistics)
    int* edge_src; int* edge_dst;
    int64_t cap;
    int *zero_cnt, *zero_blk;             // small-batch path only: the OTHER parity's count / block-total arrays, cleared by the fill pass for the next step
    uint32_t* mask;                       // [word][n_dst]: neighbour bit masks written by the count pass, 32 keys per word
    int word_start[kMaxScales + 1];       // first mask word of every scale (scale n has ceil(n_keys_n / 32) words)
};
constexpr int kEdgeCountWord = 48;        // tile_info[48 + n]: edges of scale n of the last neighbour search (statistics)
constexpr int kNbrChunk = 1024;
constexpr int kNbrBlock = 256;

// exclusive scan of one int per thread over a 256-thread block; returns the exclusive prefix, total in *total
__device__ inline int block_exclusive_scan_256(int v, int* total) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    __syncthreads();
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wsum[i];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + x - v;
}

// Count pass: every destination tests every key of every scale (keys staged in LDS as float4: one broadcast ds_read_b128 per
// key), counts its neighbours and leaves a bit mask of them ([word][dst], coalesced).  Fill pass: walks the set bits only.
constexpr int kFillStage = 4096;      // edges of one scale a block stages in LDS (2 x 16 KB: 16 per destination; C2 has ~20 over all four scales)
template <bool FILL>
__global__ __launch_bounds__(kNbrBlock) void k_neighbors(NbrParams P) {
    __shared__ f32x4 kx[FILL ? 2 * kFillStage / 4 : kNbrChunk];      // count pass: the key chunk; fill pass: the staging run (src | dst)
    int* const st_src = reinterpret_cast<int*>(kx);
    int* const st_dst = st_src + kFillStage;
    const int d = blockIdx.x * kNbrBlock + threadIdx.x;
    const bool act = d < P.n_dst;
    float px = 0, py = 0, pz = 0;
    if (!FILL && act) { px = P.qpos[3 * d]; py = P.qpos[3 * d + 1]; pz = P.qpos[3 * d + 2]; }
    // Fill pass: every block sums the block totals of the count pass for itself (a few hundred integers per scale) -> the total of
    // every scale and the sum of the blocks before it; block 0 also writes the tile table the following kernels read.
    __shared__ int s_tot[kMaxScales], s_pre[kMaxScales];
    if (FILL) {
        if (threadIdx.x < kMaxScales) { s_tot[threadIdx.x] = 0; s_pre[threadIdx.x] = 0; }
        __syncthreads();
        for (int n = 0; n < P.n_scales; ++n) {
            int tot = 0, pre = 0;
            for (int i = threadIdx.x; i < (int)gridDim.x; i += kNbrBlock) {
                const int v = P.blk[(size_t)n * gridDim.x + i];
                tot += v;
                if (i < (int)blockIdx.x) pre += v;
            }
            for (int o = 32; o >= 1; o >>= 1) { tot += __shfl_xor(tot, o); pre += __shfl_xor(pre, o); }
            if ((threadIdx.x & 63) == 0) { atomicAdd(&s_tot[n], tot); atomicAdd(&s_pre[n], pre); }      // integers: order-independent
        }
        __syncthreads();
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            int tiles = 0;
            int64_t edges = 0;
            P.tile_info[0] = 0; P.tile_info[16] = 0;
            for (int n = 0; n < P.n_scales; ++n) {
                edges += s_tot[n];
                tiles += (s_tot[n] + 31) / 32;
                P.tile_info[n + 1] = tiles;
                P.tile_info[16 + n + 1] = (int)min(edges, (int64_t)0x7fffffff);
                P.tile_info[kEdgeCountWord + n] = s_tot[n];
            }
            const int ovf = edges > P.cap ? 1 : 0;
            P.tile_info[40] = ovf;
            if (ovf) { P.tile_info[kFlagOverflow] = 1; P.tile_info[kFlagOverflowSeen] = 1; }      // sticky within one API call / until the host has seen it
            if (P.edge_hist) *P.edge_hist += edges;
            if (ovf) for (int n = 0; n <= P.n_scales; ++n) P.tile_info[n] = 0;     // no tiles: downstream kernels do nothing
        }
    }
    int64_t scale_base = 0;
    for (int n = 0; n < P.n_scales; ++n) {
        const int s0 = P.scale_start[n], s1 = P.scale_start[n + 1];
        const int w0 = P.word_start[n], nw = P.word_start[n + 1] - w0;
        if (FILL) {
            const int mine = act ? P.cnt[(size_t)n * P.n_dst + d] : 0;
            if (P.zero_cnt != nullptr && blockIdx.y == 0) {
                if (act) P.zero_cnt[(size_t)n * P.n_dst + d] = 0;
                if (blockIdx.x == 0) for (int i = threadIdx.x; i < (int)gridDim.x; i += kNbrBlock) P.zero_blk[(size_t)n * gridDim.x + i] = 0;
            }
            int total;
            const int ex = block_exclusive_scan_256(mine, &total);
            const int o = s_pre[n] + ex;
            const int64_t base = scale_base + o;
            const int64_t blk_base = scale_base + s_pre[n];      // first edge of this block's destinations in the scale
            scale_base += s_tot[n];
            if (act && blockIdx.y == 0) P.off[(size_t)n * P.n_dst + d] = o;
            // Small batches (gridDim.y = G > 1): a handful of blocks would walk ~150 divergent bit-loop iterations each while the chip idles (16 of
            // the pass's 23.5 us at 16 poses: timing builds, profiles/r05zc_fill_timing.txt).  The mask WORDS are split over G blocks per destination
            // block: every one of them repeats the cheap part (counts, scans, the popcounts of the words before its own) and walks only its share.
            if (gridDim.y > 1) {
                if (!act) continue;
                const int nW = P.word_start[P.n_scales];
                const int gw0 = (int)((int64_t)blockIdx.y * nW / gridDim.y), gw1 = (int)((int64_t)(blockIdx.y + 1) * nW / gridDim.y);
                if (w0 >= gw1 || w0 + nw <= gw0) continue;          // none of this scale's words are mine
                int c = 0;
                for (int g = 0; g < nw && w0 + g < gw1; ++g) {
                    uint32_t word = P.mask[(size_t)(w0 + g) * P.n_dst + d];
                    if (w0 + g < gw0) { c += __builtin_popcount(word); continue; }
                    while (word) {
                        const int bit = __builtin_ctz(word);
                        word &= word - 1;
                        if (base + c < P.cap) { P.edge_src[base + c] = s0 + 32 * g + bit; P.edge_dst[base + c] = d; }
                        ++c;
                    }
                }
                continue;
            }
            // The edges of the block's 256 destinations are ONE contiguous run of the list.  Written straight from the bit walk every store
            // instruction scatters 64 lanes over 64 cache lines (~20 edges apart) and the walk issues two of them per iteration.  So the walk fills a
            // staging run in LDS (ex = this destination's offset inside the block) and the block copies it out with coalesced stores (C2: 38.6 ->
            // 30.9 us, profiles/r05zd_fill_timing.txt); a block whose run does not fit writes directly.
            if (total <= kFillStage) {
                if (act) {
                    int c = ex;
                    for (int g = 0; g < nw; ++g) {
                        uint32_t word = P.mask[(size_t)(w0 + g) * P.n_dst + d];
                        while (word) {
                            const int bit = __builtin_ctz(word);
                            word &= word - 1;
                            st_src[c] = s0 + 32 * g + bit; st_dst[c] = d;
                            ++c;
                        }
                    }
                }
                __syncthreads();
                for (int i = threadIdx.x; i < total; i += kNbrBlock)
                    if (blk_base + i < P.cap) { P.edge_src[blk_base + i] = st_src[i]; P.edge_dst[blk_base + i] = st_dst[i]; }
                __syncthreads();
                continue;
            }
            if (!act) continue;
            int c = 0;
            for (int g = 0; g < nw; ++g) {
                uint32_t word = P.mask[(size_t)(w0 + g) * P.n_dst + d];
                while (word) {
                    const int bit = __builtin_ctz(word);
                    word &= word - 1;
                    if (base + c < P.cap) { P.edge_src[base + c] = s0 + 32 * g + bit; P.edge_dst[base + c] = d; }
                    ++c;
                }
            }
        } else {
            const float r2 = P.r2[n];
            int c = 0;
            for (int c0 = s0; c0 < s1; c0 += kNbrChunk) {      // kNbrChunk is a multiple of 32: words never straddle chunks
                const int nc = min(kNbrChunk, s1 - c0);
                __syncthreads();
                for (int i = threadIdx.x; i < nc; i += kNbrBlock) {
                    const float* kp = P.key_x + (size_t)(c0 + i) * 3;
                    kx[i] = f32x4{kp[0], kp[1], kp[2], 0.0f};
                }
                __syncthreads();
                if (!act) continue;
                for (int i0 = 0; i0 < nc; i0 += 32) {
                    uint32_t word = 0;
                    const int ni = min(32, nc - i0);
                    if (ni == 32 && r2 > 0.0f && c + 32 <= P.max_neighbors) {
                        // common case (full word, finite radius, cap cannot bind): two keys per iteration on packed fp32 ops
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll 4
                        for (int i = 0; i < 32; i += 2) {
                            const f32x4 ka = kx[i0 + i], kb = kx[i0 + i + 1];
                            const f32x2_t dx = f32x2_t{ka[0], kb[0]} - px, dy = f32x2_t{ka[1], kb[1]} - py, dz = f32x2_t{ka[2], kb[2]} - pz;
                            const f32x2_t d2 = dx * dx + dy * dy + dz * dz;
                            word |= (d2[0] < r2 ? 1u : 0u) << i;
                            word |= (d2[1] < r2 ? 2u : 0u) << i;
                        }
                        c += __builtin_popcount(word);
                    } else {
                        for (int i = 0; i < ni; ++i) {
                            const f32x4 k = kx[i0 + i];
                            const float dx = k[0] - px, dy = k[1] - py, dz = k[2] - pz;
                            const float d2 = dx * dx + dy * dy + dz * dz;
                            const bool in = (r2 <= 0.0f) || (d2 < r2);
                            if (in && (r2 <= 0.0f || c < P.max_neighbors)) { word |= 1u << i; ++c; }
                        }
                    }
                    P.mask[(size_t)(w0 + (c0 - s0 + i0) / 32) * P.n_dst + d] = word;
                }
            }
            if (act) P.cnt[(size_t)n * P.n_dst + d] = c;
            int total;
            (void)block_exclusive_scan_256(act ? c : 0, &total);
            if (threadIdx.x == 0) P.blk[(size_t)n * gridDim.x + blockIdx.x] = total;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Small batches (the deployment regime: 10-20 poses x 650-900 steps, reference evaluate_real_mug.ipynb:188-190, configs/panda_mug/server.yaml:2).
// With a few thousand destination nodes the thread-per-destination count pass above is a serial walk over all keys on a handful of CUs.  Here
// the count pass is spread over (32-key word, destination) pairs -- every thread builds ONE mask word, 32 distance tests -- and also carries
// the pose preparation (each thread transforms its own query point; one more grid row builds the Wigner-D records): one short launch instead
// of two.  Counts and block totals accumulate with integer atomics (order-independent) into arrays that alternate between two sets from step
// to step: the fill pass (k_neighbors<true>, unchanged) clears the set the NEXT step will accumulate into.  Same masks, same edge order
// (scale, dst, src) as the two-pass search; requires that the neighbour cap cannot bind (keys of a finite scale <= max_neighbors: the host
// checks) and N_d <= kNbrSmallMax.
constexpr int kNbrSmallMax = 32768;
template <int L>
__global__ __launch_bounds__(kNbrBlock) void k_nbr_masks_small(NbrParams P, const float* __restrict__ Ts, const double* __restrict__ Ts64,
                                                               const float* __restrict__ qx, int nQ, int nT, float* __restrict__ pose, float* __restrict__ qpos) {
    const int d = blockIdx.x * kNbrBlock + threadIdx.x, w = blockIdx.y;
    const int n_words = P.word_start[P.n_scales];
    if (w == n_words) {                 // the extra grid row: one thread per pose builds its record
        if (d < nT) {
            float T[7];
            load_pose(Ts, Ts64, d, T);
            pose_record<L>(T, pose + (size_t)d * pose_rec<L>());
        }
        return;
    }
    int n = 0;
    while (w >= P.word_start[n + 1]) ++n;
    const int g = w - P.word_start[n], k0 = P.scale_start[n] + 32 * g, ni = min(32, P.scale_start[n + 1] - k0);
    __shared__ f32x4 kx[32];
    __shared__ int wtot[kNbrBlock / 64];
    if (threadIdx.x < ni) { const float* kp = P.key_x + (size_t)(k0 + threadIdx.x) * 3; kx[threadIdx.x] = f32x4{kp[0], kp[1], kp[2], 0.0f}; }
    __syncthreads();
    int c = 0;
    if (d < P.n_dst) {
        const int t = d / nQ, q = d - t * nQ;
        float T[7], px, py, pz;
        load_pose(Ts, Ts64, t, T);
        pose_apply(T, qx[3 * q], qx[3 * q + 1], qx[3 * q + 2], px, py, pz);
        if (w == 0) { float* o = qpos + (size_t)d * 3; o[0] = px; o[1] = py; o[2] = pz; }
        const float r2 = P.r2[n];
        uint32_t word = 0;
        for (int i = 0; i < ni; ++i) {
            const f32x4 k = kx[i];
            const float dx = k[0] - px, dy = k[1] - py, dz = k[2] - pz;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if ((r2 <= 0.0f) || (d2 < r2)) word |= 1u << i;
        }
        P.mask[(size_t)w * P.n_dst + d] = word;
        c = __builtin_popcount(word);
        if (c) atomicAdd(P.cnt + (size_t)n * P.n_dst + d, c);
    }
    int tot = c;                        // this block's share of the fill pass's block total (same 256 destinations per block)
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        tot = 0;
        for (int i = 0; i < kNbrBlock / 64; ++i) tot += wtot[i];
        if (tot) atomicAdd(P.blk + (size_t)n * gridDim.x + blockIdx.x, tot);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Explicit edge lists (UNet layers: the graphs come from dedf_fps / dedf_radius, int64 like torch_cluster's, sorted by destination):
// 32-bit copies for the edge kernel, per-destination counts / offsets for k_aggregate, the tile table, and a validity check
// (sortedness, index ranges) whose result lands in tile_info[43].
constexpr int kFlagBadEdges = 43;
__global__ void k_edge_lists(const int64_t* __restrict__ src64, const int64_t* __restrict__ dst64, int64_t n_edges, int n_src, int n_dst,
                             int* __restrict__ esrc, int* __restrict__ edst, int* __restrict__ cnt, int* __restrict__ off, int* __restrict__ tile_info) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        tile_info[0] = 0; tile_info[1] = (int)((n_edges + 31) / 32);
        tile_info[16] = 0; tile_info[17] = (int)n_edges; tile_info[40] = 0;
    }
    if (i < n_edges) {
        const int64_t s = src64[i], d = dst64[i];
        bool bad = s < 0 || s >= n_src || d < 0 || d >= n_dst;
        if (i > 0 && dst64[i - 1] > d) bad = true;
        if (bad) tile_info[kFlagBadEdges] = 1;
        esrc[i] = bad ? 0 : (int)s; edst[i] = bad ? 0 : (int)d;
    }
    if (i < n_dst) {      // lower bounds of i and i + 1 in the sorted destination list
        auto lb = [&](int64_t v) { int64_t lo = 0, hi = n_edges; while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (dst64[m] < v) lo = m + 1; else hi = m; } return lo; };
        const int64_t a = lb(i), b = lb(i + 1);
        off[i] = (int)a; cnt[i] = (int)(b - a);
    }
}

// Edge-workspace sizing (dedf_set_key_clouds): how many key points of its own scale lie within that scale's radius of a key point, summed per
// scale -- the degree a query point ON the scene surface would see, which is where the denoised poses end up.  One thread per key point.
__global__ void k_self_degree(const float* __restrict__ key_x, int n_keys, int n_scales, const int* __restrict__ scale_start, const float* __restrict__ r2,
                              int max_neighbors, unsigned long long* __restrict__ deg_sum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    int n = 0;
    while (n + 1 < n_scales && i >= scale_start[n + 1]) ++n;
    const int a = scale_start[n], b = scale_start[n + 1];
    int c = b - a;
    if (r2[n] > 0.0f) {
        const float x = key_x[3 * i], y = key_x[3 * i + 1], z = key_x[3 * i + 2];
        c = 0;
        for (int j = a; j < b; ++j) {
            const float dx = key_x[3 * j] - x, dy = key_x[3 * j + 1] - y, dz = key_x[3 * j + 2] - z;
            c += dx * dx + dy * dy + dz * dz < r2[n] ? 1 : 0;
        }
        c = min(c, max_neighbors);
    }
    atomicAdd(deg_sum + n, (unsigned long long)c);
}

__global__ void k_or_flag(const int* __restrict__ flag, int* __restrict__ sticky) {
    if (*flag) *sticky = 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// Joint softmax over ALL scales' edges of one destination node + weighted aggregation (graph_attention.py:253-266;
// scatter_logsumexp / scatter restated: max-shifted, empty segments give 0).  One wave per destination, lane = float4 of
// the value record; deterministic (fixed edge order, no atomics).
// TILE: edges per tile of the edge kernel that wrote the records
template <int L, int TILE = 32>
__global__ void k_aggregate(const float* __restrict__ edge_out, const int* __restrict__ cnt, const int* __restrict__ off,
                            const int* __restrict__ tile_info, int n_dst, int n_scales, float* __restrict__ z) {
    constexpr int D = feat_dim<L>(), REC = edge_rec<L>(), NV = D / 4;
    const int lane = threadIdx.x & 63;
    // wave-uniform destination index: its counts / offsets / record addresses then live in scalar registers (scalar loads)
    const int d = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (d >= n_dst) return;
    if (tile_info[40]) return;
    // lane p-th float4 of the record: channels ci[p] .. ci[p] + 3 (lmax 3: 88 float4 per record, two per lane)
    constexpr int NP = (NV + 63) / 64;
    int ci[NP], head[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        ci[p] = 4 * (lane + 64 * p);
        int l = 0;
        while (l < L && ci[p] >= blk_off(l + 1)) ++l;
        head[p] = ((ci[p] - blk_off(l)) % mul_of(l)) / (mul_of(l) / kHeads);
    }
    // k_edge left one record per (destination, 32-edge tile) segment at the segment's first edge: the destination's first edge
    // and every tile boundary (scale-relative multiples of 32) inside its edge range.  All scales' counts and offsets are
    // requested first, then the records of all scales form ONE list that is walked U at a time.
    int first[kMaxScales], fb[kMaxScales], pre[kMaxScales + 1];       // all wave-uniform
    pre[0] = 0;
#pragma unroll
    for (int n = 0; n < kMaxScales; ++n) {
        first[n] = 0; fb[n] = 0;
        int nr = 0;
        if (n < n_scales) {
            const int c = cnt[(size_t)n * n_dst + d];
            const int o = off[(size_t)n * n_dst + d];
            first[n] = tile_info[16 + n] + o;        // edge index of the destination's first record
            fb[n] = TILE - (o & (TILE - 1));         // distance (in edges) to the next tile boundary
            nr = c == 0 ? 0 : 1 + (fb[n] < c ? (c - fb[n] + TILE - 1) / TILE : 0);
        }
        pre[n + 1] = pre[n] + nr;
    }
    const int total = pre[kMaxScales];
    // single pass, online softmax: running max / sum per head, accumulator rescaled when a head's max grows
    float mx[kHeads], sum[kHeads] = {0, 0, 0, 0};
    for (int h = 0; h < kHeads; ++h) mx[h] = -INFINITY;
    f32x4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = f32x4{0, 0, 0, 0};
#ifndef DEDF_AGG_U
#define DEDF_AGG_U 4
#endif
    constexpr int U = DEDF_AGG_U;          // records per iteration: all their loads are in flight together (2 / 8: see profiles/r05zl_aggregate_u.txt)
    for (int j0 = 0; j0 < total; j0 += 64) {
        // lane k holds the edge index of record j0 + k of the flattened list
        const int k = j0 + lane;
        int ek = 0;
#pragma unroll
        for (int n = 0; n < kMaxScales; ++n)
            if (k >= pre[n] && k < pre[n + 1]) { const int jj = k - pre[n]; ek = first[n] + (jj == 0 ? 0 : fb[n] + TILE * (jj - 1)); }
        const int jend = min(64, total - j0);
        for (int j = 0; j < jend; j += U) {
            f32x4 lg[U], v[U][NP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lg[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int p = 0; p < NP; ++p) v[u][p] = f32x4{0, 0, 0, 0};
                if (j + u < jend) {
                    const float* r = edge_out + (size_t)__builtin_amdgcn_readlane(ek, j + u) * REC;
                    lg[u] = ld4(r + D);
#pragma unroll
                    for (int p = 0; p < NP; ++p) if (ci[p] < D) v[u][p] = ld4(r + ci[p]);
                }
            }
            float sc[kHeads], p[U][kHeads];
            for (int h = 0; h < kHeads; ++h) {
                float m_new = mx[h];
#pragma unroll
                for (int u = 0; u < U; ++u) m_new = fmaxf(m_new, lg[u][h]);
                sc[h] = expf(mx[h] - m_new);            // exp(-inf) = 0 on the first group
                float ps = 0.0f;
#pragma unroll
                for (int u = 0; u < U; ++u) { p[u][h] = expf(lg[u][h] - m_new); ps += p[u][h]; }
                sum[h] = sum[h] * sc[h] + ps;
                mx[h] = m_new;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int hd = head[q];
                const float sh = hd == 0 ? sc[0] : (hd == 1 ? sc[1] : (hd == 2 ? sc[2] : sc[3]));
                acc[q] = acc[q] * sh;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float ph = hd == 0 ? p[u][0] : (hd == 1 ? p[u][1] : (hd == 2 ? p[u][2] : p[u][3]));
                    acc[q] = acc[q] + v[u][q] * ph;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) if (ci[q] < D) {
        const int hd = head[q];
        const float s = hd == 0 ? sum[0] : (hd == 1 ? sum[1] : (hd == 2 ? sum[2] : sum[3]));
        const float inv = s > 0.0f ? 1.0f / s : 0.0f;
        st4(z + (size_t)d * D + ci[q], acc[q] * inv);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_pose_reduce(const float* __restrict__ node_out, int nT, int nQ, float* __restrict__ ang, float* __restrict__ lin,
                                                    int* __restrict__ flags, const float* __restrict__ node_spin = nullptr) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= nT) return;
    float s[6] = {0, 0, 0, 0, 0, 0};
    for (int q = lane; q < nQ; q += 64) {
        const float* o = node_out + ((size_t)t * nQ + q) * 8;
        const f32x4 a = ld4(o), b = ld4(o + 4);
        if (node_spin != nullptr) {      // (the node kernel ran split: NodeParams::split)
            const f32x4 sp = ld4(node_spin + ((size_t)t * nQ + q) * 4);
            s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3] + sp[0]; s[4] += b[0] + sp[1]; s[5] += b[1] + sp[2];
        } else {
        s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3]; s[4] += b[0]; s[5] += b[1];
        }
    }
    for (int m = 32; m >= 1; m >>= 1)
        for (int i = 0; i < 6; ++i) s[i] += __shfl_xor(s[i], m, 64);
    if (lane == 0) {
        reduce_status(flags, s);
        lin[3 * t] = s[0]; lin[3 * t + 1] = s[1]; lin[3 * t + 2] = s[2];
        ang[3 * t] = s[3]; ang[3 * t + 1] = s[4]; ang[3 * t + 2] = s[5];
    }
}

// Sampler: sum over the query points of one pose (k_pose_reduce's order, bit for bit) and the Langevin update of that pose in one
// launch — one wave per pose, lane 0 carries the float64 update.
__global__ __launch_bounds__(64) void k_reduce_langevin(const float* __restrict__ node_out, int nQ, float* __restrict__ ang_out,
                                                        float* __restrict__ lin_out, LangevinParams P, int* __restrict__ flags,
                                                        const float* __restrict__ node_spin = nullptr) {
    const int t = blockIdx.x;
    if (t >= P.nT) return;
    reduce_pose_and_step(node_out, nQ, t, (int)threadIdx.x, ang_out, lin_out, P, flags, node_spin);
}

}  // namespace dedf
