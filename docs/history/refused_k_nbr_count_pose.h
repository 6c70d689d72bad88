// Round 6, measured and refused (profiles/r06j_nbr_cull_ab2.log: count pass 65 us against 48 us for the thread-per-destination pass over all keys;
// with k_pose_prep's launch folded in the step breaks even): kept here as text, not compiled.
// ------------------------------------------------------------------------------------------------------------------------
// Count pass with POSE-ALIGNED blocks and key culling (round 6).  k_neighbors<false> above is bound by LDS reads: every wave reads every key
// (1 024 broadcast ds_read_b128 per wave and step at C2).  The query points of ONE pose sit within a few centimetres of each other, so a block
// that holds the destinations of one pose (chunks of 256 query points) first culls the keys of a finite scale against the bounding box of its
// points grown by the radius -- 256 threads, 4 keys each, compacted IN KEY ORDER -- and every destination then tests the survivors only
// (C2: ~200 of the 1 024 keys).  Same masks, counts and (scale, dst, src) order as the pass it replaces (the cull is conservative: the exact
// fp32 distance test decides; the neighbour cap counts hits in key order as before); the per-256-destination block totals the fill pass sums are
// accumulated with integer atomics (order-independent; the host clears them).  The block also does the pose preparation of its pose --
// every thread transforms its own query point, the grid's first blocks build the Wigner-D records -- so k_pose_prep's launch goes too.
template <int L>
__global__ __launch_bounds__(kNbrBlock) void k_nbr_count_pose(NbrParams P, const float* __restrict__ Ts, const double* __restrict__ Ts64, const float* __restrict__ qx,
                                                             int nQ, int n_chunks, int n_poses, float* __restrict__ pose, float* __restrict__ qpos, int n_fill_blocks) {
    __shared__ f32x4 s_ck[kNbrChunk];                         // survivors of the current key chunk: coordinates, .w = index inside the chunk (as int bits)
    __shared__ uint32_t s_mask[kNbrChunk / 32][kNbrBlock];    // this chunk's mask words, [word][thread]
    __shared__ float s_box[2][kNbrBlock / 64][3];
    __shared__ int s_wsum[kNbrBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the first ceil(nT / 256) blocks build the pose records, one thread per pose (the Wigner-D construction is a long serial chain: beside the
    // count blocks it costs nothing, inside one it would hold the block's LDS for its whole length)
    const int n_rec = (n_poses + kNbrBlock - 1) / kNbrBlock;
    if ((int)blockIdx.x < n_rec) {
        const int tp = blockIdx.x * kNbrBlock + tid;
        if (tp < n_poses) {
            float Tp[7];
            load_pose(Ts, Ts64, tp, Tp);
            pose_record<L>(Tp, pose + (size_t)tp * pose_rec<L>());
        }
        return;
    }
    const int bid = blockIdx.x - n_rec;
    const int t = bid / n_chunks, chunk = bid - t * n_chunks;
    const int q = chunk * kNbrBlock + tid;
    const bool act = q < nQ;
    const int d = t * nQ + (act ? q : 0);
    float T[7];
    load_pose(Ts, Ts64, t, T);
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (act) {
        pose_apply(T, qx[3 * q], qx[3 * q + 1], qx[3 * q + 2], px, py, pz);
        float* o = qpos + (size_t)d * 3;
        o[0] = px; o[1] = py; o[2] = pz;
    }
    // bounding box of the block's points
    float lo[3] = {act ? px : INFINITY, act ? py : INFINITY, act ? pz : INFINITY}, hi[3] = {act ? px : -INFINITY, act ? py : -INFINITY, act ? pz : -INFINITY};
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o >= 1; o >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64)); }
        if (lane == 0) { s_box[0][wave][k] = lo[k]; s_box[1][wave][k] = hi[k]; }
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k)
        for (int w = 0; w < kNbrBlock / 64; ++w) { lo[k] = fminf(lo[k], s_box[0][w][k]); hi[k] = fmaxf(hi[k], s_box[1][w][k]); }
    const int fb0 = (t * nQ + chunk * kNbrBlock) / kNbrBlock;      // the 256-destination fill block of this block's first destination
    for (int n = 0; n < P.n_scales; ++n) {
        const int s0 = P.scale_start[n], s1 = P.scale_start[n + 1], w0 = P.word_start[n];
        const float r2 = P.r2[n];
        const float r2c = r2 * (1.0f + 1e-5f) + 1e-6f;             // the cull's bound: above every distance the exact test can accept
        int c = 0;
        for (int c0 = s0; c0 < s1; c0 += kNbrChunk) {
            const int nc = min(kNbrChunk, s1 - c0), nwc = (nc + 31) / 32;
            __syncthreads();                                       // (the previous chunk's survivors and words have been consumed)
            for (int g = 0; g < nwc; ++g) s_mask[g][tid] = 0u;
            int ncand;
            if (r2 > 0.0f) {
                // cull: thread tid looks at keys 4 tid .. 4 tid + 3 of the chunk (key order = thread order = compaction order)
                f32x4 kk[4];
                int keep = 0;
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * tid + j;
                    bool in = false;
                    if (i < nc) {
                        const float* kp = P.key_x + (size_t)(c0 + i) * 3;
                        const float kx_ = kp[0], ky_ = kp[1], kz_ = kp[2];
                        const float ex = fmaxf(fmaxf(lo[0] - kx_, kx_ - hi[0]), 0.0f), ey = fmaxf(fmaxf(lo[1] - ky_, ky_ - hi[1]), 0.0f), ez = fmaxf(fmaxf(lo[2] - kz_, kz_ - hi[2]), 0.0f);
                        in = ex * ex + ey * ey + ez * ez < r2c;
                        kk[j] = f32x4{kx_, ky_, kz_, __int_as_float(i)};
                    }
                    keep |= in ? (1 << j) : 0;
                }
                const int mine = __builtin_popcount(keep);
                int x = mine;
                for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
                if (lane == 63) s_wsum[wave] = x;
                __syncthreads();
                int base = x - mine;
                for (int w = 0; w < wave; ++w) base += s_wsum[w];
                ncand = 0;
                for (int w = 0; w < kNbrBlock / 64; ++w) ncand += s_wsum[w];
                for (int j = 0; j < 4; ++j) if (keep & (1 << j)) s_ck[base++] = kk[j];
                __syncthreads();
                if (act) {
                    for (int i = 0; i < ncand; ++i) {
                        const f32x4 k = s_ck[i];
                        const float dx = k[0] - px, dy = k[1] - py, dz = k[2] - pz;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 < r2 && c < P.max_neighbors) {
                            const int ki = __float_as_int(k[3]);
                            s_mask[ki >> 5][tid] |= 1u << (ki & 31);
                            ++c;
                        }
                    }
                }
            } else if (act) {                                      // all pairs: every key of the chunk
                for (int g = 0; g < nwc; ++g) { const int ni = min(32, nc - 32 * g); s_mask[g][tid] = ni == 32 ? 0xffffffffu : ((1u << ni) - 1u); }
                c += nc;
            }
            if (act) for (int g = 0; g < nwc; ++g) P.mask[(size_t)(w0 + (c0 - s0) / 32 + g) * P.n_dst + d] = s_mask[g][tid];
        }
        if (act) P.cnt[(size_t)n * P.n_dst + d] = c;
        // block totals of the fill pass (256 consecutive destinations each): a wave's destinations touch at most two of them
        const int fb = act ? d / kNbrBlock : -1, fb_first = __shfl(fb, 0, 64);
        int ca = (act && fb == fb_first) ? c : 0, cb = (act && fb != fb_first) ? c : 0;
        for (int o = 32; o >= 1; o >>= 1) { ca += __shfl_xor(ca, o, 64); cb += __shfl_xor(cb, o, 64); }
        if (lane == 0 && fb_first >= 0) {
            if (ca) atomicAdd(P.blk + (size_t)n * n_fill_blocks + fb_first, ca);
            if (cb) atomicAdd(P.blk + (size_t)n * n_fill_blocks + fb_first + 1, cb);
        }
        (void)fb0;
    }
}

