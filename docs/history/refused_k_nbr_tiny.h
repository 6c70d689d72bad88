// ------------------------------------------------------------------------------------------------------------------------
// Tiny batches -- the reference's own pick deployment: 10-20 poses x the TWO static keypoints of pick_lowres (configs/panda_mug/pick_lowres/
// score_model_configs.yaml:76-80, evaluate_real_mug.ipynb:188-190), 40 destination nodes.  The whole neighbour search of a step -- pose records,
// transformed query points, masks, counts, offsets, tile table, edge lists -- is ONE workgroup's work there: one launch instead of two, no
// global atomics, no alternating count sets (round 6: 22 -> XX us of a 106 us step).  Same masks, same edge order (scale, dst, src) and the same
// cnt / off / tile_info contents as the other paths; like k_nbr_masks_small it requires that the neighbour cap cannot bind (host-checked).
constexpr int kNbrTinyDst = 128, kNbrTinyWords = 64, kNbrTinyKeys = 2048, kNbrTinyBlock = 1024;
template <int L>
__global__ __launch_bounds__(kNbrTinyBlock) void k_nbr_tiny(NbrParams P, const float* __restrict__ Ts, const double* __restrict__ Ts64,
                                                            const float* __restrict__ qx, int nQ, int nT, float* __restrict__ pose, float* __restrict__ qpos) {
    __shared__ f32x4 s_key[kNbrTinyKeys];
    __shared__ uint32_t s_mask[kNbrTinyWords][kNbrTinyDst];
    __shared__ int s_cnt[kMaxScales][kNbrTinyDst], s_off[kMaxScales][kNbrTinyDst];
    __shared__ float s_pos[kNbrTinyDst][3];
    __shared__ int s_tot[kMaxScales];
    const int tid = threadIdx.x, Nd = P.n_dst, ns = P.n_scales, nW = P.word_start[ns];
    for (int i = tid; i < P.n_keys; i += kNbrTinyBlock) { const float* kp = P.key_x + (size_t)i * 3; s_key[i] = f32x4{kp[0], kp[1], kp[2], 0.0f}; }
    for (int i = tid; i < kMaxScales * kNbrTinyDst; i += kNbrTinyBlock) (&s_cnt[0][0])[i] = 0;
    if (tid < Nd) {
        const int t = tid / nQ, q = tid - t * nQ;
        float T[7], px, py, pz;
        load_pose(Ts, Ts64, t, T);
        pose_apply(T, qx[3 * q], qx[3 * q + 1], qx[3 * q + 2], px, py, pz);
        s_pos[tid][0] = px; s_pos[tid][1] = py; s_pos[tid][2] = pz;
        float* o = qpos + (size_t)tid * 3;
        o[0] = px; o[1] = py; o[2] = pz;
    }
    {   // the pose records (Wigner-D: the slow serial part) on the block's LAST threads, beside the mask work of the first ones
        const int tp = kNbrTinyBlock - 1 - tid;
        if (tp < nT) {
            float T[7];
            load_pose(Ts, Ts64, tp, T);
            pose_record<L>(T, pose + (size_t)tp * pose_rec<L>());
        }
    }
    __syncthreads();
    auto scale_of = [&](int w) { int n = 0; while (w >= P.word_start[n + 1]) ++n; return n; };
    for (int p = tid; p < Nd * nW; p += kNbrTinyBlock) {
        const int w = p / Nd, d = p - w * Nd, n = scale_of(w);
        const int k0 = P.scale_start[n] + 32 * (w - P.word_start[n]), ni = min(32, P.scale_start[n + 1] - k0);
        const float r2 = P.r2[n], px = s_pos[d][0], py = s_pos[d][1], pz = s_pos[d][2];
        uint32_t word = 0;
        for (int i = 0; i < ni; ++i) {
            const f32x4 k = s_key[k0 + i];
            const float dx = k[0] - px, dy = k[1] - py, dz = k[2] - pz;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if ((r2 <= 0.0f) || (d2 < r2)) word |= 1u << i;
        }
        s_mask[w][d] = word;
        const int c = __builtin_popcount(word);
        if (c) atomicAdd(&s_cnt[n][d], c);
    }
    __syncthreads();
    {   // wave n: exclusive prefix of scale n's counts over the destinations (two per lane)
        const int wave = tid >> 6, lane = tid & 63;
        if (wave < ns) {
            const int a = 2 * lane < Nd ? s_cnt[wave][2 * lane] : 0, b = 2 * lane + 1 < Nd ? s_cnt[wave][2 * lane + 1] : 0;
            int x = a + b;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
            s_off[wave][2 * lane] = x - a - b; s_off[wave][2 * lane + 1] = x - b;
            if (lane == 63) s_tot[wave] = x;
        }
    }
    __syncthreads();
    int ebase[kMaxScales + 1];
    ebase[0] = 0;
    for (int n = 0; n < kMaxScales; ++n) ebase[n + 1] = ebase[n] + (n < ns ? s_tot[n] : 0);
    const bool ovf = (int64_t)ebase[ns] > P.cap;
    if (tid == 0) {
        int tiles = 0;
        P.tile_info[0] = 0; P.tile_info[16] = 0;
        for (int n = 0; n < ns; ++n) {
            tiles += (s_tot[n] + 31) / 32;
            P.tile_info[n + 1] = ovf ? 0 : tiles;
            P.tile_info[16 + n + 1] = ebase[n + 1];
            P.tile_info[kEdgeCountWord + n] = s_tot[n];
        }
        P.tile_info[40] = ovf ? 1 : 0;
        if (ovf) { P.tile_info[kFlagOverflow] = 1; P.tile_info[kFlagOverflowSeen] = 1; }
        if (P.edge_hist) *P.edge_hist += ebase[ns];
    }
    for (int i = tid; i < ns * Nd; i += kNbrTinyBlock) {
        const int n = i / Nd, d = i - n * Nd;
        P.cnt[(size_t)n * Nd + d] = s_cnt[n][d];
        P.off[(size_t)n * Nd + d] = s_off[n][d];
    }
    if (ovf) return;
    for (int p = tid; p < Nd * nW; p += kNbrTinyBlock) {
        const int w = p / Nd, d = p - w * Nd, n = scale_of(w), w0 = P.word_start[n];
        uint32_t word = s_mask[w][d];
        if (!word) continue;
        int c = ebase[n] + s_off[n][d];
        for (int g = w0; g < w; ++g) c += __builtin_popcount(s_mask[g][d]);
        const int s0 = P.scale_start[n] + 32 * (w - w0);
        while (word) {
            const int bit = __builtin_ctz(word);
            word &= word - 1;
            P.edge_src[c] = s0 + bit; P.edge_dst[c] = d;
            ++c;
        }
    }
}

