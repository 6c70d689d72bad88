// libdedf.so — host side of the C ABI declared in include/dedf.h (see there for the reference interfaces replaced).
// Build: __graft_entry__.build() (dedf_api.hip + dedf_kernels.hip x kKernelUnits, compiled in parallel), or in one piece:
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -shared -fPIC -DDEDF_SINGLE_TU dedf_api.hip -o libdedf.so   (gfx950 only, no fallbacks)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../../include/dedf.h"
#include "dedf_misc.h"
#include "dedf_pack.h"

using namespace dedf;

#include "dedf_kernels.h"
#include "dedf_graph.h"
#if !defined(DEDF_SINGLE_TU)
#define DEDF_DECL(unit, ...) extern template __global__ __VA_ARGS__;
DEDF_KERNEL_LIST(DEDF_DECL)
#undef DEDF_DECL
#endif
__global__ void k_energy_reduce(const float* __restrict__ node_out, int nT, int nQ, float* __restrict__ energy, int* __restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nT) return;
    float s = 0.0f;
    for (int q = 0; q < nQ; ++q) s += node_out[((size_t)t * nQ + q) * 8];
    if (flags[kFlagOverflow]) s = __builtin_nanf("");              // see dedf_misc.h::reduce_status
    else if (!(fabsf(s) <= 3.0e38f)) { flags[kFlagNonFinite] = 1; flags[kFlagNonFiniteSeen] = 1; }
    energy[t] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool ensure(size_t n) {
        if (n <= bytes) return true;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, n) != hipSuccess) return false;
        bytes = n;
        return true;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct dedf_handle {
    dedf_config cfg{};
    std::string err;
    bool host_only = false;
    int L = 2;
    ParamSpec spec, kspec;                // canonical parameter list in the reference's TRUE shapes (the C ABI's blob) / in the kernels' shapes
    std::vector<float> params, kparams;   // (they differ at lmax 3 only: 8x3e runs zero-padded as 16x3e, dedf_pack.h::pad_params)
    DevBuf d_qf_true;                     // lmax 3: staging of the true-shape query features before they are padded
    Image edge_img, node_img;
    EdgeOffsets eo{};
    NodeOffsets no{};
    int n_cu = 256;
    // device: weights
    DevBuf d_edge_w, d_node_w, d_nat;     // d_nat: natural-layout weights for the small kernels
    size_t nat_tw1 = 0, nat_tb1 = 0, nat_tw2 = 0, nat_tb2 = 0, nat_wpre = 0, nat_bpre = 0, nat_lnw = 0, nat_lnb = 0, nat_wsrc = 0, nat_bsrc = 0, nat_brows = 0, nat_wdst = 0, nat_bdst = 0;
    size_t nat_qw1 = 0, nat_qb1 = 0, nat_qw2 = 0, nat_qb2 = 0, nat_qlnw = 0, nat_qlnb = 0, nat_qwdst = 0, nat_qbdst = 0, nat_qwskip = 0, nat_qbskip = 0;      // query_time_encoding
    DevBuf d_msg_dst;
    // device: scene / query
    DevBuf d_key_x, d_key_f, d_msg, d_qx, d_qf, d_qw;
    int n_keys = 0, nQ = 0;
    int scale_start[kMaxScales + 1] = {0};
    bool have_keys = false, have_query = false;
    // device: per call
    DevBuf d_key_w; bool have_key_w = false;              // key-point attention weights (use_src_point_attn)
    const LangevinParams* fused_step = nullptr;           // dedf_sample: the per-pose reduction also carries this step's Langevin update
    DevBuf d_qrows, d_qrows_steps;                        // query_time_encoding: time rows (dedf_misc.h::k_time_query) of the poses of a call / of ALL steps of dedf_sample
    DevBuf d_tb_steps; const float* tb_step = nullptr;    // dedf_sample: the time-bias rows of ALL steps are computed up front; tb_step = this step's rows
    std::vector<float> h_tsteps;
    // Pinned host block of dedf_sample (round 5: per-call overhead): [0, 64) ints = the tile_info block of the finished call, [64, 72) = the radial
    // table's guard words, [128, ...) = the step times on their way up.  The status words come down with ONE asynchronous copy in front of the call's
    // final synchronisation (they were two blocking copies behind it), and dedf_get_stats right after a dedf_sample reads them from here.
    int* h_pin = nullptr; size_t h_pin_floats = 0;
    bool stats_fresh = false;         // h_pin holds the status of the last finished call and nothing was enqueued since
    int radial_table = 1;         // sampler: tabulate the radial network's front per launch (DEDF_RADIAL_TABLE=0 / dedf_set_radial_table turn it off)
    DevBuf d_rtab, d_rtab_err;    // table rows; per-scale accuracy words (largest |interpolated - exact| activation at the interval midpoints)
    float rtab_err_bound = 1e-5f; // a scale whose word exceeds it evaluates its front per edge (DEDF_RADIAL_TABLE_BOUND)
    // dedf_sample, round 4, OPT-IN (DEDF_RTAB_ASYNC=1) -- built, correct (the sampler tests pass on it), and SLOWER: the tables of the coming steps are
    // generated on a SIDE stream into a ring of kRtabRing tables while the main stream runs the current step (a step's table depends on its time
    // only), which takes the 34 us generator off the step's dependent chain and lets batches below kRtabMinNodes use the table.  Measured
    // (profiles/r04j_*): C2 341.4 k -> 330.5 k pose-steps/s (k_edge 2.39 -> 2.48 ms: the generator's 1 217 workgroups take wave slots from the
    // persistent edge grid, whose displaced waves then run a round of their own), 16 poses 0.156 -> 0.197 ms per step (the generator competes with
    // the single-round, latency-bound kernels of a small step).  The default generates every table on the main stream, as rounds 2-3 did.
    static constexpr int kRtabRing = 4;
    hipStream_t side = nullptr;
    hipEvent_t rt_tab[kRtabRing] = {nullptr, nullptr, nullptr, nullptr}, rt_used[kRtabRing] = {nullptr, nullptr, nullptr, nullptr}, rt_ready = nullptr;
    DevBuf d_rtab_ring;
    int rtab_fin = kRtabFinite, rtab_inf = kRtabInfinite;      // grid intervals per finite / all-pairs scale (DEDF_RTAB_FIN / DEDF_RTAB_INF: experiments)
    int tab_slot = -1;            // >= 0 inside dedf_sample's loop: the ring slot that holds THIS step's table
    bool rtab_async = false;      // DEDF_RTAB_ASYNC=1 turns the side-stream generation on (A/B)
    DevBuf d_cnt2, d_blk2; int small_parity = 0; int64_t small_layout = -1;      // the two alternating count sets of the small-batch neighbour path
    // Verdict of the last dedf_score / dedf_energy call (they never synchronise): the status words are copied to pinned host memory behind the
    // call's kernels; the NEXT entry point of this handle looks at them (check_pending) and fails if the call overflowed its edge workspace or
    // produced a non-finite result -- so that a caller who never reads dedf_get_stats does not keep working with NaN scores.
    int* h_flags = nullptr;           // pinned, 64 ints (the tile_info block)
    hipEvent_t ev_flags = nullptr;
    bool flags_pending = false;
    bool clear_seen = false;          // the host has consumed a sticky verdict (kFlagOverflowSeen / kFlagNonFiniteSeen): clear the words at the next call
    bool so2 = false;                 // the edge kernels of this handle run both depth-wise TPs in the edge-aligned frame (dedf_edge.h: SO2; its own
                                      // packed image, dedf_pack.h::pack_edge<L, true>).  DEDF_SO2=0 keeps the general form (A/B, tests)
    bool small_batch_path = true;     // N_d <= kNbrSmallMax (32 768): word-parallel neighbour masks + fused pose preparation (DEDF_SMALL_BATCH=0 turns it off: A/B, tests)
    bool defer_check = false;         // dedf_layer_defer_check
    DevBuf d_sticky;
    dedf_handle* ws_owner = nullptr;  // dedf_layer_share_workspace: the UNet-layer handle whose per-call workspace (messages, edge lists, segment
                                      // records, aggregate, flags) this layer uses instead of its own
    std::vector<dedf_handle*> ws_borrowers;   // ... and, on the owner's side, the handles that currently borrow it (detached by dedf_destroy(owner))
    bool want_field = false;          // dedf_field: the node kernel also writes the field / emb of every node
    DevBuf d_nspin;                   // [N_d][4]: the ang_spin half of the node kernel's output when it runs split (NodeParams::split)
    DevBuf d_Ts, d_time, d_tb, d_pose, d_qpos, d_cnt, d_off, d_blk, d_tile, d_esrc, d_edst, d_eout, d_z, d_nout, d_ang, d_lin, d_T64, d_dbgw, d_dbge, d_dbgf, d_dbgo, d_mask;
    int64_t edge_cap = 0;
    double est_degree = 0.0;          // dedf_set_key_clouds: sum over the scales of the key points' mean self-degree (k_self_degree) = the edges a query
                                      // point on the scene surface has; sizes the automatic edge workspace
    int sample_retries = 0;           // repeats of the last dedf_sample call (dedf_stats)
    int64_t auto_per_dst = 96;        // automatic edge workspace: edges per destination node (grown by dedf_sample when a call overflowed)
    DevBuf d_deg;
    int last_nT = 0;
    bool debug = false;
    hipStream_t last_stream = nullptr;
    // live profiling (bench.py roofline)
    bool profile = false;
    std::vector<hipEvent_t> ev;           // 7 events per evaluation
    size_t ev_used = 0;
    int64_t prof_evals = 0, prof_dst = 0;
    DevBuf d_hist, d_phase;
    ~dedf_handle() {
        for (auto e : ev) (void)hipEventDestroy(e);
        for (int i = 0; i < kRtabRing; ++i) { if (rt_tab[i]) (void)hipEventDestroy(rt_tab[i]); if (rt_used[i]) (void)hipEventDestroy(rt_used[i]); }
        if (rt_ready) (void)hipEventDestroy(rt_ready);
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

// Entry points run on the handle's device and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define DEDF_ON_DEVICE(h)                                   \
    DeviceGuard dev_guard__((h)->cfg.device);              \
    if (!dev_guard__.ok) return fail(h, DEDF_ERR_RUNTIME, "hipSetDevice failed")
// clears the sticky status words (overflow, non-finite) at the start of an API call
#define DEDF_CLEAR_FLAGS(h, st)                                                                                                              \
    do {                                                                                                                                     \
        HIPCK(h, hipMemsetAsync((h)->d_tile.as<int>() + kFlagOverflow, 0, (kFlagTimeVaries + 1 - kFlagOverflow) * sizeof(int), st));          \
        if ((h)->clear_seen) { HIPCK(h, hipMemsetAsync((h)->d_tile.as<int>() + kFlagOverflowSeen, 0, 2 * sizeof(int), st)); (h)->clear_seen = false; } \
    } while (0)

int fail(dedf_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
#define HIPCK(h, call)                                                                                          \
    do {                                                                                                        \
        hipError_t e__ = (call);                                                                                \
        if (e__ != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// dedf_score / dedf_energy enqueue and return; this hands their verdict to the host without a synchronisation of its own
int post_flags(dedf_handle* h, hipStream_t st) {
    if (!h->h_flags) {
        HIPCK(h, hipHostMalloc((void**)&h->h_flags, 64 * sizeof(int), hipHostMallocDefault));
        HIPCK(h, hipEventCreateWithFlags(&h->ev_flags, hipEventDisableTiming));
    }
    HIPCK(h, hipMemcpyAsync(h->h_flags, h->d_tile.p, 64 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCK(h, hipEventRecord(h->ev_flags, st));
    h->flags_pending = true;
    return DEDF_OK;
}
// block = false: only if the previous call has already completed (an entry point that does not synchronise stays that way)
int check_pending(dedf_handle* h, bool block) {
    if (!h->flags_pending) return DEDF_OK;
    if (block) HIPCK(h, hipEventSynchronize(h->ev_flags));
    else if (hipEventQuery(h->ev_flags) != hipSuccess) return DEDF_OK;
    h->flags_pending = false;
    // (the "seen" words are sticky on the device until consumed here: a verdict of ANY call since the last look, not only of the latest one)
    const bool ovf = (h->h_flags[40] | h->h_flags[kFlagOverflow] | h->h_flags[kFlagOverflowSeen]) != 0, nonfin = (h->h_flags[kFlagNonFinite] | h->h_flags[kFlagNonFiniteSeen]) != 0;
    if (ovf || nonfin) h->clear_seen = true;      // the next call of the handle clears them on its stream, in front of its kernels
    if (ovf)
        return fail(h, DEDF_ERR_RUNTIME, "a previous dedf_score / dedf_energy call (since this handle's status was last looked at) overflowed its edge workspace: its outputs are NaN (dedf_config.max_edges, or let "
                                         "dedf_sample grow the automatic workspace); reported by the next call because dedf_score never synchronises");
    if (nonfin)
        return fail(h, DEDF_ERR_RUNTIME, "a previous dedf_score / dedf_energy call (since this handle's status was last looked at) produced a non-finite result (an operand left the fp16 window of the split GEMMs, "
                                         "or the inputs were not finite): its outputs are NaN; reported by the next call because dedf_score never synchronises");
    return DEDF_OK;
}

int check_config(const dedf_config* c, std::string& why) {
    if (!c) { why = "null config"; return DEDF_ERR_INVALID; }
    if (c->unet_layer) {       // one {radial, gnn} layer of the UNet feature extractor (unet_feature_extractor.py:141-202)
        if ((c->lmax != 2 && c->lmax != 3) || c->mul[0] != 64 || c->mul[1] != 32 || c->mul[2] != 16 || (c->lmax == 3 && c->mul[3] != 16)) {
            why = "UNet layer: irreps must be the kernel shapes 64x0e+32x1e+16x2e (lmax 2) or 64x0e+32x1e+16x2e+16x3e (lmax 3: 8x3e zero-padded; narrower levels run zero-padded too, see unet_valid)"; return DEDF_ERR_UNSUPPORTED; }
        if (c->num_heads != kHeads) { why = "num_heads must be 4"; return DEDF_ERR_UNSUPPORTED; }
        if (c->fc_neurons[0] != 64 || c->fc_neurons[1] != 32 || c->fc_neurons[2] != 32) { why = "UNet layer: fc_neurons must be [64, 32, 32]"; return DEDF_ERR_UNSUPPORTED; }
        if (c->irreps_mlp_mid != kMlpMid) { why = "irreps_mlp_mid must be 3"; return DEDF_ERR_UNSUPPORTED; }
        if (c->n_scales != 1 || !(c->radii[0] > 0)) { why = "UNet layer: n_scales = 1 and radii[0] = the level's connection radius"; return DEDF_ERR_INVALID; }
        if (c->ebm || c->use_src_point_attn) { why = "UNet layer: ebm / use_src_point_attn do not apply"; return DEDF_ERR_UNSUPPORTED; }
        for (int l = 0; l <= c->lmax; ++l)
            if (c->unet_valid[l] < 0 || c->unet_valid[l] > c->mul[l] || (c->unet_valid[l] > 0 && c->unet_valid[l] % 4)) { why = "UNet layer: unet_valid[l] must be 0 or a multiple of 4 up to mul[l]"; return DEDF_ERR_INVALID; }
        if (c->lmax == 3 && (c->unet_valid[3] == 0 || c->unet_valid[3] > 8)) { why = "UNet layer, lmax 3: unet_valid[3] must name the true 3e multiplicity (8 or 4)"; return DEDF_ERR_INVALID; }
        for (int l = 0; l < 3; ++l)
            if (c->unet_fc_valid[l] < 0 || c->unet_fc_valid[l] > c->fc_neurons[l]) { why = "UNet layer: unet_fc_valid out of range"; return DEDF_ERR_INVALID; }
        if (c->unet_narrow)
            for (int l = 0; l <= c->lmax; ++l)
                if (c->unet_valid[l] != (32 >> l)) { why = "UNet layer: unet_narrow needs unet_valid = {32, 16, 8 (, 4)}"; return DEDF_ERR_INVALID; }
        return DEDF_OK;
    }
    if (c->lmax < 1 || c->lmax > 3) { why = "lmax must be 1, 2 or 3"; return DEDF_ERR_UNSUPPORTED; }
    for (int l = 0; l <= c->lmax; ++l)
        if (c->mul[l] != true_mul(l)) { why = "irreps must be 64x0e+32x1e(+16x2e(+8x3e))"; return DEDF_ERR_UNSUPPORTED; }
    if (c->lmax == 3) {      // instantiated at lmax 3: the score head and the EBM critic / context-free field of the panda shapes, full and half precision
        const bool ok3 = c->fc_neurons[1] == kFc1 && c->fc_neurons[2] == kFc2 && (c->ebm || c->time_emb_mlp[2] == 64);
        const bool kp3 = c->ebm && c->fc_neurons[1] == 32 && c->fc_neurons[2] == 32;       // KeypointExtractor fields
        if (!ok3 && !kp3) { why = "lmax 3 is instantiated for fc_neurons [64 + 64, 128, 64] (score head), [64, 128, 64] / [64, 32, 32] (EBM critic / context-free field)"; return DEDF_ERR_UNSUPPORTED; }
    }
    if (c->num_heads != kHeads) { why = "num_heads must be 4"; return DEDF_ERR_UNSUPPORTED; }
    const bool t_small = c->time_emb_mlp[0] == 256 && c->time_emb_mlp[1] == 128 && c->time_emb_mlp[2] == 64;
    const bool t_big = c->time_emb_mlp[0] == 512 && c->time_emb_mlp[1] == 256 && c->time_emb_mlp[2] == 128;   // sapien high-res configs
    if (!t_small && !t_big) { why = "time_emb_mlp must be [256,128,64] or [512,256,128]"; return DEDF_ERR_UNSUPPORTED; }
    const bool mlp_wide = c->fc_neurons[1] == kFc1 && c->fc_neurons[2] == kFc2;
    const bool mlp_narrow = c->fc_neurons[1] == 32 && c->fc_neurons[2] == 32;       // sapien place_* score heads
    // (a score head WITHOUT edge_time_encoding -- the reference constructor's default, score_head.py:40-41 -- has the 64-wide pre-linear of the EBM
    //  head and takes its time through query_time_encoding alone: instantiated for fc_neurons [64, 128, 64], full precision)
    const bool no_edge_time = !c->ebm && c->fc_neurons[0] == kLenEmb && c->query_time_encoding;
    if ((c->fc_neurons[0] != (c->ebm ? kLenEmb : kLenEmb + c->time_emb_mlp[2]) && !no_edge_time) || !(mlp_wide || mlp_narrow)) {
        why = "fc_neurons must resolve to [64 + time_emb, 128, 64] or [64 + time_emb, 32, 32] (score head with edge_time_encoding), [64, 128, 64] (score head with query_time_encoding only), or [64,128,64] / [64,32,32] (EBM head / context-free field, no time encoding)"; return DEDF_ERR_UNSUPPORTED; }
    if (no_edge_time && (!mlp_wide || c->half_gemm)) {
        why = "the score head without edge_time_encoding is instantiated for fc_neurons [64, 128, 64] in full precision"; return DEDF_ERR_UNSUPPORTED; }
    if (mlp_narrow && c->ebm && c->lmax < 2) {
        why = "the context-free field with the 32-wide radial MLP (KeypointExtractor) is instantiated for lmax 2 and 3"; return DEDF_ERR_UNSUPPORTED; }
    if (mlp_narrow && !c->ebm && c->fc_neurons[0] != 128) {
        why = "the 32-wide radial MLP is instantiated for the score head with the 128-wide pre-linear only"; return DEDF_ERR_UNSUPPORTED; }
    if (c->length_emb_dim != kLenEmb) { why = "length_emb_dim must be 64"; return DEDF_ERR_UNSUPPORTED; }
    if (c->irreps_mlp_mid != kMlpMid) { why = "irreps_mlp_mid must be 3"; return DEDF_ERR_UNSUPPORTED; }
    if (c->half_gemm && c->lmax == 1 && c->fc_neurons[0] == 192) { why = "lmax 1 with a 128-channel time embedding is not instantiated"; return DEDF_ERR_UNSUPPORTED; }
    if (c->query_time_encoding) {
        const bool ok2 = c->lmax == 2 && ((c->fc_neurons[0] == 128 && (mlp_wide || mlp_narrow)) || (c->fc_neurons[0] == 192 && mlp_wide) || no_edge_time);
        const bool ok3 = c->lmax == 3 && (c->fc_neurons[0] == 128 || no_edge_time) && mlp_wide;
        const bool ok1 = c->lmax == 1 && (c->fc_neurons[0] == 128 || no_edge_time) && mlp_wide;
        const bool okh = !c->half_gemm || (c->lmax == 2 && c->fc_neurons[0] == 128 && mlp_wide);
        if (c->ebm || !(ok1 || ok2 || ok3) || !okh) {
            why = "query_time_encoding is instantiated for the score head: lmax 2 with fc_neurons [128,128,64] (also in half precision) / [128,32,32] / [192,128,64], lmax 1 and 3 with [128,128,64]; without edge_time_encoding [64,128,64] at lmax 1-3"; return DEDF_ERR_UNSUPPORTED; }
    }
    if (c->n_scales < 1 || c->n_scales > kMaxScales) { why = "n_scales out of range"; return DEDF_ERR_INVALID; }
    bool inf = false;
    for (int n = 0; n < c->n_scales; ++n) {
        if (c->radii[n] <= 0) inf = true;
        else if (inf) { why = "Finite cluster radius cannot come after infinite cluster radius"; return DEDF_ERR_INVALID; }
    }
    if (c->max_neighbors <= 0) { why = "max_neighbors must be positive"; return DEDF_ERR_INVALID; }
    return DEDF_OK;
}

// the run-time irreps bookkeeping of the packers' schema must be what the kernels' compile-time description says
template <int L> bool irreps_consistent(const IrrepsRT& K) {
    bool ok = K.L == L && K.dtp_wn == dtp_wn<L>() && K.stp_wn == stp_wn<L>() && K.lin0_rows() == lin0_rows<L>() && K.f1_rows0() == f1_rows0<L>() &&
              (int)K.dtp.size() == dtp_num_paths<L>() && (int)K.stp.size() == stp_num_paths<L>() && K.stp_k[0] == stp_k<L>(0) && K.stp_k[1] == stp_k<L>(1);
    for (int l = 0; l <= L; ++l) ok = ok && K.mul[l] == mul_of(l) && K.hid[l] == hid_of(l) && K.dtp_k[l] == dtp_k<L>(l);
    for (int p = 0; ok && p < dtp_num_paths<L>(); ++p) ok = K.dtp[p].wstart == dtp_path<L>(p).wstart && K.dtp[p].kofs == dtp_path<L>(p).kofs && K.dtp[p].l3 == dtp_path<L>(p).l3;
    for (int p = 0; ok && p < stp_num_paths<L>(); ++p) ok = K.stp[p].wstart == stp_path<L>(p).wstart && K.stp[p].kofs == stp_path<L>(p).kofs;
    return ok;
}
// shapes with an edge-aligned-frame instantiation (dedf_kernel_list.h): the full-precision lmax-2 score head with the [128, 128, 64] radial network
bool so2_instantiated(const dedf_config& c) {
    if (c.unet_layer)       // UNet layers (round 5): lmax 2 / 3 with the [64, 32, 32] radial network
        return c.lmax >= 2 && c.fc_neurons[0] == 64 && c.fc_neurons[1] == 32 && c.fc_neurons[2] == 32;
    const int F0 = c.fc_neurons[0];
    const bool wide = c.fc_neurons[1] == 128 && c.fc_neurons[2] == 64, narrow = c.fc_neurons[1] == 32 && c.fc_neurons[2] == 32;
    if (c.lmax == 3) return (F0 == 128 && wide) || (F0 == 64 && (wide || narrow));
    if (c.lmax == 2) return ((F0 == 128 || F0 == 64) && (wide || narrow)) || (F0 == 192 && wide);
    return ((F0 == 128 || F0 == 64) && wide) || (F0 == 128 && narrow);
}
// the GENERAL form of the handle's full-precision edge kernels is instantiated too (so that DEDF_SO2=0 can select it): the headline shapes only
bool general_instantiated(const dedf_config& c) {
    if (c.half_gemm || c.unet_layer || c.query_time_encoding) return false;
    return (c.lmax == 2 || c.lmax == 3) && c.fc_neurons[0] == 128 && c.fc_neurons[1] == 128 && c.fc_neurons[2] == 64;
}
template <int L> void pack_all(dedf_handle* h) {
    bool done = false;
    if (h->so2) { pack_edge<L, true>(h->cfg, h->kspec, h->kparams.data(), h->edge_img, h->eo); done = true; }
    if (!done) pack_edge<L>(h->cfg, h->kspec, h->kparams.data(), h->edge_img, h->eo);
    pack_node<L>(h->cfg, h->kspec, h->kparams.data(), h->node_img, h->no, h->cfg.unet_layer ? 0.0f : h->eo.est_value);
    if (getenv("DEDF_SCALE_DEBUG"))
        fprintf(stderr, "dedf scales: s3 %d (msg x %g) su %d est_value %.3g | node bz %g bn %g bh %g %g %g bf %g bt %g %g\n", h->eo.s3, h->eo.msg_scale, h->eo.su, h->eo.est_value, h->no.sc.bz, h->no.sc.bn,
                h->no.sc.bh[0], h->no.sc.bh[1], h->no.sc.bh[2], h->no.sc.bf, h->no.sc.bt[0], h->no.sc.bt[1]);
}

// the C ABI's canonical parameter list: TRUE shapes for the score / critic heads; UNet-layer handles take the kernel shapes (the caller pads)
ParamSpec spec_for(const dedf_config* c) {
    return build_spec(IrrepsRT(c->lmax, c->unet_layer != 0), *c);
}

size_t feat_dim_rt(int L) { return L == 1 ? feat_dim<1>() : (L == 2 ? feat_dim<2>() : feat_dim<3>()); }            // kernel layout
size_t true_feat_dim_rt(int L) { return L == 1 ? true_feat_dim<1>() : (L == 2 ? true_feat_dim<2>() : true_feat_dim<3>()); }
size_t edge_rec_rt(int L) { return feat_dim_rt(L) + kHeads; }
size_t dtp_wn_rt(int L) { return L == 1 ? dtp_wn<1>() : (L == 2 ? dtp_wn<2>() : dtp_wn<3>()); }
size_t pose_rec_rt(int L) { return L >= 3 ? pose_rec<3>() : pose_rec<2>(); }

int upload_weights(dedf_handle* h) {
    const dedf_config& c = h->cfg;
    const ParamSpec& S = h->spec;          // natural-layout weights of the small kernels: the TRUE shapes (k_src_message<L, ., TRUE_IN>)
    const float* B = h->params.data();
    const IrrepsRT T(h->L, c.unet_layer != 0);
    if (!h->d_edge_w.ensure(h->edge_img.data.size() * 4) || !h->d_node_w.ensure(h->node_img.data.size() * 4))
        return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(weights) failed");
    HIPCK(h, hipMemcpy(h->d_edge_w.p, h->edge_img.data.data(), h->edge_img.data.size() * 4, hipMemcpyHostToDevice));
    HIPCK(h, hipMemcpy(h->d_node_w.p, h->node_img.data.data(), h->node_img.data.size() * 4, hipMemcpyHostToDevice));
    // natural-layout weights for the small kernels
    std::vector<float> nat;
    auto put = [&](const float* p, size_t n) { size_t o = nat.size(); nat.insert(nat.end(), p, p + n); return o; };
    const int ns = c.n_scales;
    if (c.unet_layer) {        // linear_src (no bias) and linear_dst (+ bias), block.py:109-115
        const size_t sq = T.sq();
        h->nat_wsrc = put(S.get(B, "gnn.linear_src.tp.weight"), sq);
        h->nat_wdst = put(S.get(B, "gnn.linear_dst.tp.weight"), sq);
        h->nat_bdst = put(S.get(B, "gnn.linear_dst.bias.0"), mul_of(0));
        if (!h->d_nat.ensure(nat.size() * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(nat) failed");
        HIPCK(h, hipMemcpy(h->d_nat.p, nat.data(), nat.size() * 4, hipMemcpyHostToDevice));
        return DEDF_OK;
    }
    h->nat_tw1 = nat.size();
    const size_t tE = c.time_emb_mlp[0], tH = c.time_emb_mlp[1], tT = c.time_emb_mlp[2], F0 = c.fc_neurons[0];
    for (int n = 0; n < ns; ++n) put(S.get(B, "time_mlps_multiscale." + std::to_string(n) + ".0.weight"), tH * tE);
    h->nat_tb1 = nat.size();
    for (int n = 0; n < ns; ++n) put(S.get(B, "time_mlps_multiscale." + std::to_string(n) + ".0.bias"), tH);
    h->nat_tw2 = nat.size();
    for (int n = 0; n < ns; ++n) put(S.get(B, "time_mlps_multiscale." + std::to_string(n) + ".2.weight"), tT * tH);
    h->nat_tb2 = nat.size();
    for (int n = 0; n < ns; ++n) put(S.get(B, "time_mlps_multiscale." + std::to_string(n) + ".2.bias"), tT);
    h->nat_wpre = nat.size();
    for (int n = 0; n < ns; ++n) put(S.get(B, "key_tensor_field.edge_scalars_pre_linears." + std::to_string(n) + ".0.weight"), F0 * F0);
    h->nat_bpre = nat.size();
    for (int n = 0; n < ns; ++n) put(S.get(B, "key_tensor_field.edge_scalars_pre_linears." + std::to_string(n) + ".0.bias"), F0);
    const std::string blk = "key_tensor_field.gnn_block_init";
    const size_t nirr = T.sum_mul(), sq = T.sq();
    h->nat_lnw = put(S.get(B, blk + ".prenorm_src.affine_weight"), nirr);
    h->nat_lnb = put(S.get(B, blk + ".prenorm_src.affine_bias"), mul_of(0));
    h->nat_wsrc = put(S.get(B, blk + ".linear_src.tp.weight"), sq);
    if (!c.query_time_encoding) h->nat_bsrc = put(S.get(B, blk + ".linear_src.bias.0"), mul_of(0));
    else {      // use_dst_feature: linear_src has no bias (gnn_block.py:127); the query-side time MLP and the three small matrices the time rows go through
        const std::vector<float> zero(mul_of(0), 0.0f);
        h->nat_bsrc = put(zero.data(), zero.size());
        h->nat_qw1 = put(S.get(B, "query_time_mlp.0.weight"), tH * tE); h->nat_qb1 = put(S.get(B, "query_time_mlp.0.bias"), tH);
        h->nat_qw2 = put(S.get(B, "query_time_mlp.2.weight"), tT * tH); h->nat_qb2 = put(S.get(B, "query_time_mlp.2.bias"), tT);
        h->nat_qlnw = put(S.get(B, blk + ".prenorm_dst.affine_weight"), tT); h->nat_qlnb = put(S.get(B, blk + ".prenorm_dst.affine_bias"), tT);
        h->nat_qwdst = put(S.get(B, blk + ".linear_dst.tp.weight"), tT * mul_of(0)); h->nat_qbdst = put(S.get(B, blk + ".linear_dst.bias.0"), mul_of(0));
        h->nat_qwskip = put(S.get(B, blk + ".skip_1.skip.tp.weight"), tT * mul_of(0)); h->nat_qbskip = put(S.get(B, blk + ".skip_1.skip.bias.0"), mul_of(0));
    }
    if (c.fc_neurons[0] == kLenEmb) {   // no edge time encoding (EBM head, score head with query_time_encoding only): the pre-linear "time rows" are just its bias, row-packed once
        std::vector<float> rows;
        for (int n = 0; n < ns; ++n) {
            const float* b = S.get(B, "key_tensor_field.edge_scalars_pre_linears." + std::to_string(n) + ".0.bias");
            auto r = pack_rows(c.fc_neurons[0], [&](int i) { return b[i]; });
            rows.insert(rows.end(), r.begin(), r.end());
        }
        h->nat_brows = put(rows.data(), rows.size());
    }
    if (!h->d_nat.ensure(nat.size() * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(nat) failed");
    HIPCK(h, hipMemcpy(h->d_nat.p, nat.data(), nat.size() * 4, hipMemcpyHostToDevice));
    return DEDF_OK;
}

int ensure_workspace(dedf_handle* h, int nT) {
    const int L = h->L;
    const size_t D = feat_dim_rt(L), REC = edge_rec_rt(L), WN = dtp_wn_rt(L);
    const size_t Nd = (size_t)nT * h->nQ;
    const int ns = h->cfg.n_scales;
    // worst case: every key of every scale is a neighbour (capped per scale by max_neighbors for finite scales)
    int64_t per_dst = 0;
    for (int n = 0; n < ns; ++n) {
        const int64_t k = h->scale_start[n + 1] - h->scale_start[n];
        per_dst += h->cfg.radii[n] > 0 ? std::min<int64_t>(k, h->cfg.max_neighbors) : k;
    }
    int64_t cap = (int64_t)Nd * per_dst;
    if (h->cfg.max_edges > 0) cap = std::min(cap, h->cfg.max_edges);
    else cap = std::min(cap, std::max<int64_t>((int64_t)Nd * h->auto_per_dst, 1 << 20));    // auto: 96 edges per destination node, or 1.5 x the degree of a query
                                                                                            // point on the scene surface where that is more (dedf_set_key_clouds)
    cap = std::min<int64_t>(cap, 0x7fffffff - 64);
    cap = std::max<int64_t>(cap, 64);
    h->edge_cap = cap;
    const bool new_tile = h->d_tile.p == nullptr;      // (the status block starts from zero: its "seen" words are only ever cleared once consumed)
    bool ok = h->d_Ts.ensure((size_t)nT * 7 * 4) && h->d_time.ensure((size_t)nT * 4) && h->d_tb.ensure((size_t)nT * ns * 256 * 4) &&
              h->d_pose.ensure((size_t)nT * pose_rec_rt(L) * 4) && h->d_qpos.ensure(Nd * 3 * 4) && h->d_cnt.ensure(Nd * ns * 4) &&
              h->d_off.ensure(Nd * ns * 4) && h->d_blk.ensure(((Nd + kNbrBlock - 1) / kNbrBlock) * ns * 4 + 64) && h->d_tile.ensure(64 * 4) && h->d_esrc.ensure((size_t)cap * 4) &&
              h->d_edst.ensure((size_t)cap * 4) && h->d_eout.ensure((size_t)cap * REC * 4) && h->d_z.ensure(Nd * D * 4) &&
              h->d_nout.ensure(Nd * 8 * 4) && h->d_ang.ensure((size_t)nT * 3 * 4) && h->d_lin.ensure((size_t)nT * 3 * 4) &&
              h->d_T64.ensure((size_t)nT * 7 * 8);
    {   // neighbour bit masks of the count pass: one word per 32 keys of every scale, per destination
        size_t words = 0;
        for (int n = 0; n < ns; ++n) words += (size_t)(h->scale_start[n + 1] - h->scale_start[n] + 31) / 32;
        ok = ok && h->d_mask.ensure(words * Nd * 4);
    }
    if (ok && new_tile) ok = hipMemset(h->d_tile.p, 0, 64 * 4) == hipSuccess;
    if (ok && h->debug) ok = h->d_dbgw.ensure((size_t)cap * WN * 4);
    if (ok && h->cfg.query_time_encoding) ok = h->d_qrows.ensure((size_t)nT * kQueryTimeRow * 4);
    if (!ok) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(workspace) failed");
    if (h->d_eout.bytes >= (1ull << 32) || h->d_z.bytes >= (1ull << 32))
        ;   // buffer descriptors address 4 GiB: z is read through one (see launch), eout is written with flat stores
    return DEDF_OK;
}

// Time embedding -> pre-linear bias rows tb[row][scale][F0] for `rows` times read at time[row * time_stride]
void launch_time_bias(dedf_handle* h, hipStream_t st, const float* time, int time_stride, int rows, float* tb, int* varies = nullptr) {
    const dedf_config& c = h->cfg;
    const float* nat = h->d_nat.as<float>();
    TimeParams tp{};
    tp.time = time; tp.time_stride = time_stride;
    tp.w1 = nat + h->nat_tw1; tp.b1 = nat + h->nat_tb1; tp.w2 = nat + h->nat_tw2; tp.b2 = nat + h->nat_tb2;
    tp.wpre = nat + h->nat_wpre; tp.bpre = nat + h->nat_bpre;
    tp.E = c.time_emb_mlp[0]; tp.H = c.time_emb_mlp[1]; tp.TE = c.time_emb_mlp[2];
    tp.max_time = c.max_time; tp.time_enc_n = c.time_enc_n; tp.tb = tb; tp.varies = varies;
    hipLaunchKernelGGL(k_time_bias, dim3(rows, c.n_scales), dim3(256), 0, st, tp);
}

// query_time_encoding: the time rows (dedf_misc.h::k_time_query) for `rows` times read at time[row * time_stride]
void launch_time_query(dedf_handle* h, hipStream_t st, const float* time, int time_stride, int rows, float* out) {
    const dedf_config& c = h->cfg;
    const float* nat = h->d_nat.as<float>();
    TimeQueryParams tp{};
    tp.time = time; tp.time_stride = time_stride;
    tp.w1 = nat + h->nat_qw1; tp.b1 = nat + h->nat_qb1; tp.w2 = nat + h->nat_qw2; tp.b2 = nat + h->nat_qb2;
    tp.ln_w = nat + h->nat_qlnw; tp.ln_b = nat + h->nat_qlnb; tp.wdst = nat + h->nat_qwdst; tp.bdst = nat + h->nat_qbdst;
    tp.wskip = nat + h->nat_qwskip; tp.bskip = nat + h->nat_qbskip;
    tp.E = c.time_emb_mlp[0]; tp.H = c.time_emb_mlp[1]; tp.TE = c.time_emb_mlp[2];
    tp.max_time = c.max_time; tp.time_enc_n = c.time_enc_n; tp.out_scale = h->eo.msg_scale; tp.rows = out;
    hipLaunchKernelGGL(k_time_query, dim3(rows), dim3(256), 0, st, tp);
}
// ... and where the current evaluation reads them: this step's row inside dedf_sample (h->tb_step names the step), else one row per pose / one shared row
const float* query_time_rows(dedf_handle* h, int time_stride, int& pose_stride) {
    if (h->tb_step != nullptr) {
        const size_t step = (size_t)(h->tb_step - h->d_tb_steps.as<float>()) / ((size_t)h->cfg.n_scales * h->cfg.fc_neurons[0]);
        pose_stride = 0;
        return h->d_qrows_steps.as<float>() + step * kQueryTimeRow;
    }
    pose_stride = time_stride ? kQueryTimeRow : 0;
    return h->d_qrows.as<float>();
}

// Persistent grids: as many waves per CU as are RESIDENT at once (4 for the 512-register kernels with <= 40 KB of LDS -- every k_edge / k_node
// since the lmax-3 edge kernels went from 52 to 39.5 KB; the occupancy query stays: it is what sized their grids at 3 and 2 before).  A grid larger than that runs in rounds and the last, partly filled round costs a whole one.
template <auto Kernel, int Cap = 4> int waves_per_cu() {
    static const int n = [] {
        int v = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, Kernel, 64, 0) != hipSuccess || v < 1) v = 4;
        return std::min(v, Cap);
    }();
    return n;
}
#define DEDF_LAUNCH_PERSISTENT(KERNEL, MAX_BLOCKS, ST, ARG) \
    hipLaunchKernelGGL((KERNEL), dim3(std::min<int>((MAX_BLOCKS), h->n_cu * std::min(edge_wpc_limit(), waves_per_cu<(KERNEL), 8>()))), dim3(64), 0, ST, ARG)
// A grid whose tile count is known on the host (the node kernel): the persistent waves run ceil(tiles / resident waves) rounds either way, so
// launch only as many waves as fill those rounds EVENLY (C2: 3 219 node tiles on 1 024 resident waves = 4 rounds, the last one a seventh full;
// 805 waves run the same 4 rounds with 3.1 instead of 4 waves per CU sharing the L1: k_node 0.292 -> 0.285 ms, profiles/r03t_node_balanced_ab.log).
// DEDF_NODE_BALANCED=0 restores the full grid (A/B).
inline int balanced_blocks(int tiles, int cap) {
    static const bool on = [] { const char* e = getenv("DEDF_NODE_BALANCED"); return !(e && atoi(e) == 0); }();
    if (!on || tiles <= cap) return tiles;
    const int rounds = (tiles + cap - 1) / cap;
    return (tiles + rounds - 1) / rounds;
}
inline int edge_wpc_limit() {      // experiments only: DEDF_EDGE_WAVES_PER_CU=1..4
    static const int wpc = [] { const char* e = getenv("DEDF_EDGE_WAVES_PER_CU"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 8 ? v : 4; }();      // (8: the two-waves-per-SIMD timing builds)
    return wpc;
}
// Shapes with an edge-aligned-frame instantiation (dedf_kernel_list.h; dedf_edge.h: SO2): the full-precision score heads / critics / context-free
// fields of lmax <= 2.  launch_edge picks it when the handle's image was packed for it (dedf_handle::so2).
template <int L, int F0, bool HP, int H1, int H2, int MODE> constexpr bool so2_shape() {
    if (HP && MODE != 0) return false;      // (half precision evaluates per edge: no table-reading instantiation)
    if (L == 3) return MODE == 1 ? (F0 == 128 && H1 == 128) : ((F0 == 128 && H1 == 128) || F0 == 64);
    if (MODE == 1) return (L == 2 && ((F0 == 128 && ((H1 == 128 && H2 == 64) || (H1 == 32 && H2 == 32))) || (F0 == 192 && H1 == 128 && H2 == 64))) ||
                          (L == 1 && F0 == 128 && H1 == 128 && H2 == 64);
    if (L == 2) return (F0 == 128 || F0 == 64) ? true : (F0 == 192 && H1 == 128);
    return (F0 == 128 || F0 == 64) && H1 == 128 ? true : (F0 == 128 && H1 == 32);      // lmax 1
}
// ... and the shapes whose GENERAL form is still instantiated: every half-precision one, and of the full-precision ones the headline shapes (A/B: DEDF_SO2=0)
template <int L, int F0, bool HP, int H1, int H2, int MODE> constexpr bool general_shape() {
    return (!so2_shape<L, F0, HP, H1, H2, MODE>()) || (!HP && (L == 2 || L == 3) && F0 == 128 && H1 == 128 && H2 == 64);
}
template <int L, int F0, bool HP, int H1, int H2, int MODE>
void launch_edge(dedf_handle* h, hipStream_t st, const EdgeParams& P) {
    constexpr int kAll = 1 << 30;
    if (h->cfg.query_time_encoding) {      // (validate_config admits only these shapes)
        constexpr bool qt = !HP && MODE <= 1 && (((L == 1 || L == 2 || L == 3) && F0 == 128 && H1 == 128 && H2 == 64) || (L == 2 && F0 == 128 && H1 == 32 && H2 == 32) ||
                                                 (L == 2 && F0 == 192 && H1 == 128 && H2 == 64) || (MODE == 0 && F0 == 64 && H1 == 128 && H2 == 64));
        constexpr bool qt_hp = HP && MODE == 0 && L == 2 && F0 == 128 && H1 == 128 && H2 == 64;
        if constexpr (qt || qt_hp) DEDF_LAUNCH_PERSISTENT((k_edge<L, F0, HP, H1, H2, false, MODE, false, true, true>), kAll, st, P);
        return;
    }
    if constexpr (so2_shape<L, F0, HP, H1, H2, MODE>()) {
        if (h->so2 || !general_shape<L, F0, HP, H1, H2, MODE>()) { DEDF_LAUNCH_PERSISTENT((k_edge<L, F0, HP, H1, H2, false, MODE, false, true>), kAll, st, P); return; }
    }
    if constexpr (general_shape<L, F0, HP, H1, H2, MODE>()) DEDF_LAUNCH_PERSISTENT((k_edge<L, F0, HP, H1, H2, false, MODE>), kAll, st, P);
}

// parameters of the fused edge kernel for the current state of the handle
template <int L, int F0>
EdgeParams edge_params(dedf_handle* h, int nT, int time_stride) {
    constexpr bool NOTIME = F0 == kLenEmb;      // no edge time encoding: constant pre-linear bias rows
    constexpr int D = feat_dim<L>();
    const dedf_config& c = h->cfg;
    const int ns = c.n_scales;
    const float* nat = h->d_nat.as<float>();
    EdgeParams P{};
    P.key_x = h->d_key_x.as<float>(); P.qpos = h->d_qpos.as<float>(); P.edge_src = h->d_esrc.as<int>(); P.edge_dst = h->d_edst.as<int>();
    P.tile_info = h->d_tile.as<int>(); P.msg = h->d_msg.as<float>(); P.msg_bytes = (uint32_t)((size_t)h->n_keys * D * 4);
    if constexpr (NOTIME) {
        P.tb = nat + h->nat_brows; P.tb_bytes = (uint32_t)((size_t)ns * F0 * 4); P.tb_pose_stride = 0;
    } else {
        P.tb = h->tb_step ? h->tb_step : h->d_tb.as<float>();
        P.tb_bytes = (uint32_t)((size_t)(time_stride && !h->tb_step ? nT : 1) * ns * F0 * 4);
        P.tb_pose_stride = time_stride && !h->tb_step ? ns * F0 : 0;
    }
    if (c.query_time_encoding) {
        int ps = 0;
        P.msg_dst = query_time_rows(h, time_stride, ps);
        P.qd_pose_stride = ps; P.msg_dst_bytes = (uint32_t)((size_t)(ps ? nT : 1) * kQueryTimeRow * 4);
    }
    P.nQ = h->nQ; P.n_scales = ns;
    for (int n = 0; n < ns; ++n) {
        P.radius[n] = c.radii[n] > 0 ? c.radii[n] : -1.0f;
        P.cut_begin[n] = (float)(0.8 * (double)c.radii[n]);
        P.cut_div[n] = (float)(1.0 * (double)c.radii[n] - 0.8 * (double)c.radii[n]);
    }
    P.ns_lo = (float)(0.2 * (double)c.r_mincut_nonscalar_sh);
    P.ns_div = (float)(1.0 * (double)c.r_mincut_nonscalar_sh - 0.2 * (double)c.r_mincut_nonscalar_sh);
    P.len_enc_max_r = c.length_enc_max_r;
    P.W = h->d_edge_w.as<float>(); P.W_bytes = (uint32_t)h->d_edge_w.bytes;
    const EdgeOffsets& o = h->eo;
    P.o_enc = o.o_enc; P.o_A_pre = o.o_A_pre; P.o_A_pre_l = o.o_A_pre_l; P.o_A_r1_l = o.o_A_r1_l; P.o_A_r2_l = o.o_A_r2_l; P.o_A_r3_l = o.o_A_r3_l; P.o_A_r1 = o.o_A_r1; P.o_b_r1 = o.o_b_r1; P.o_g_r1 = o.o_g_r1; P.o_be_r1 = o.o_be_r1;
    P.o_A_r2 = o.o_A_r2; P.o_b_r2 = o.o_b_r2; P.o_g_r2 = o.o_g_r2; P.o_be_r2 = o.o_be_r2; P.o_A_r3 = o.o_A_r3; P.o_off_r3 = o.o_off_r3;
    P.o_S_lin = o.o_S_lin; P.o_S_val = o.o_S_val;
    P.w_unscale = o.w_unscale; P.u_scale = o.u_scale;
    for (int l = 0; l < 4; ++l) { P.c_lin[l] = o.c_lin[l]; P.c_val[l] = o.c_val[l]; }
    P.o_b_r0 = o.o_b_r0; P.o_b_val0 = o.o_b_val0; P.o_alpha_dot = o.o_alpha_dot;
    P.key_w = h->cfg.use_src_point_attn ? h->d_key_w.as<float>() : nullptr;
    P.out = h->d_eout.as<float>();
    P.dbg_w = h->debug ? h->d_dbgw.as<float>() : nullptr;
    P.dbg_out = (h->debug && h->d_dbgo.ensure((size_t)h->edge_cap * edge_rec<L>() * 4)) ? h->d_dbgo.as<float>() : nullptr;
    P.phase_prof = nullptr;
#if defined(DEDF_PHASE_PROF)
    if (h->d_phase.ensure((size_t)h->n_cu * 4 * 16 * 8)) P.phase_prof = h->d_phase.as<unsigned long long>();
#endif
    return P;
}

// ---- the sampler's radial table (dedf_edge.h: EdgeParams::rtab) ------------------------------------------------------------------------
// instantiated for the score heads: lmax 2 with fc_neurons {128|192, 128, 64} and {128, 32, 32}; lmax 1 (round 6: config C1) and 3 with {128, 128, 64}; full precision
template <int L, int F0> constexpr bool has_radial_table() { return (L == 2 && (F0 == 128 || F0 == 192)) || ((L == 1 || L == 3) && F0 == 128); }
template <int L, int F0> bool table_instantiated(const dedf_handle* h) {
    const bool narrow = h->cfg.fc_neurons[1] == 32;
    return !h->cfg.half_gemm && !h->cfg.ebm && !(narrow && !(L == 2 && F0 == 128));
}
// Accuracy bound of scale n: 1e-5 (absolute, on the O(1) activations; the fp32 evaluation's own noise measures ~2e-6 there) for the finite
// scales, whose Gaussian length encoder is trainable.  The all-pairs scale's sinusoidal encoder has no trainable parameter and its
// interpolation error is ~1e-7; what the check sees there is the rounding of the fp32 ARGUMENT x = 10 len fr of sin / cos, up to
// x = 1000 * kRtabInfiniteSpan at the far end of the grid (ulp 1.2e-4 at 1 500: measured deviation 1.2-1.4e-4, the same noise the
// per-edge evaluation -- and the reference's own fp32 forward -- carries): its bound is two ulps of that argument.
float radial_table_bound(const dedf_handle* h, int n) {
    if (h->cfg.radii[n] > 0) return h->rtab_err_bound;
    const float xmax = 1000.0f * (float)kRtabInfiniteSpan;
    return std::max(h->rtab_err_bound, 2.0f * (std::nextafter(xmax, 2.0f * xmax) - xmax));
}
// grid of every scale + the device buffers (table rows, accuracy words)
int radial_table_setup(dedf_handle* h, EdgeParams& P) {
    const dedf_config& c = h->cfg;
    int row = 0;
    for (int n = 0; n < c.n_scales; ++n) {
        const bool fin = c.radii[n] > 0;
        const int G = fin ? h->rtab_fin : h->rtab_inf;
        const double span = fin ? (double)c.radii[n] : kRtabInfiniteSpan * (double)c.length_enc_max_r;
        P.rtab_row0[n] = row; P.rtab_n[n] = G;
        P.rtab_step[n] = (float)(span / G); P.rtab_inv_step[n] = (float)(G / span);
        row += G + 3;
    }
    const size_t bytes = (size_t)row * 64 * 4;
    if (!h->d_rtab.ensure(bytes)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(radial table) failed");
    if (!h->d_rtab_err.p) {
        if (!h->d_rtab_err.ensure(kMaxScales * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(radial table) failed");
        if (hipMemset(h->d_rtab_err.p, 0, kMaxScales * 4) != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, "hipMemset failed");
    }
    P.rtab = h->d_rtab.as<float>(); P.rtab_out = h->d_rtab.as<float>(); P.rtab_bytes = (uint32_t)bytes;
    P.rtab_err = h->d_rtab_err.as<unsigned>();
    for (int n = 0; n < c.n_scales; ++n) P.rtab_err_bound[n] = radial_table_bound(h, n);
    return DEDF_OK;
}
// the generator (one launch) and, when `check`, the accuracy check of what it produced (a second launch: every interval midpoint)
template <int L, int F0, int H1, int H2>
void launch_radial_table(dedf_handle* h, const EdgeParams& P, hipStream_t st, bool check) {
    int ntab = 0, nchk = 0;
    for (int n = 0; n < P.n_scales; ++n) { ntab += (P.rtab_n[n] + 3 + 31) / 32; nchk += (P.rtab_n[n] + 31) / 32; }
    hipLaunchKernelGGL((k_radial_table<L, F0, false, H1, H2>), dim3(std::min(ntab, h->n_cu * 4)), dim3(64), 0, st, P);
    if (check) hipLaunchKernelGGL((k_radial_check<L, F0, false, H1, H2>), dim3(std::min(nchk, h->n_cu * 4)), dim3(64), 0, st, P);
}
constexpr int kRtabAsyncMinNodes = 1024;      // with the generator off the step's dependent chain the table pays from ~one round of edge tiles on
size_t radial_table_rows(const dedf_handle* h) {
    size_t row = 0;
    for (int n = 0; n < h->cfg.n_scales; ++n) row += (h->cfg.radii[n] > 0 ? h->rtab_fin : h->rtab_inf) + 3;
    return row;
}
// side stream: the table of the step whose time-bias rows are h->tb_step, into ring slot `slot`
template <int L, int F0>
int radial_table_async(dedf_handle* h, int slot) {
    if constexpr (has_radial_table<L, F0>()) {
        EdgeParams P = edge_params<L, F0>(h, 1, 0);
        int rc = radial_table_setup(h, P);
        if (rc != DEDF_OK) return rc;
        P.rtab_out = h->d_rtab_ring.as<float>() + (size_t)slot * radial_table_rows(h) * 64;
        if (h->cfg.fc_neurons[1] == 32) { if constexpr (L == 2 && F0 == 128) launch_radial_table<L, F0, 32, 32>(h, P, h->side, false); }
        else launch_radial_table<L, F0, 128, 64>(h, P, h->side, false);
    }
    return DEDF_OK;
}
int radial_table_async_dispatch(dedf_handle* h, int slot) {
    const int F0 = h->cfg.fc_neurons[0];
    if (h->L == 1 && F0 == 128) return radial_table_async<1, 128>(h, slot);
    if (h->L == 2 && F0 == 128) return radial_table_async<2, 128>(h, slot);
    if (h->L == 2 && F0 == 192) return radial_table_async<2, 192>(h, slot);
    if (h->L == 3 && F0 == 128) return radial_table_async<3, 128>(h, slot);
    return DEDF_OK;
}
bool radial_table_instantiated_rt(const dedf_handle* h) {
    const int F0 = h->cfg.fc_neurons[0];
    if (h->L == 2 && F0 == 128) return table_instantiated<2, 128>(h);
    if (h->L == 1 && F0 == 128) return table_instantiated<1, 128>(h);
    if (h->L == 2 && F0 == 192) return table_instantiated<2, 192>(h);
    if (h->L == 3 && F0 == 128) return table_instantiated<3, 128>(h);
    return false;
}
// dedf_sample, before its loop: table + check at the time-bias rows h->tb_step (the accuracy words accumulate over the calls)
template <int L, int F0>
int radial_table_check(dedf_handle* h, hipStream_t st) {
    if constexpr (has_radial_table<L, F0>()) {
        if (!table_instantiated<L, F0>(h)) return DEDF_OK;
        EdgeParams P = edge_params<L, F0>(h, 1, 0);
        int rc = radial_table_setup(h, P);
        if (rc != DEDF_OK) return rc;
        if (h->cfg.fc_neurons[1] == 32) { if constexpr (L == 2 && F0 == 128) launch_radial_table<L, F0, 32, 32>(h, P, st, true); }
        else launch_radial_table<L, F0, 128, 64>(h, P, st, true);
    }
    return DEDF_OK;
}
int radial_table_check_dispatch(dedf_handle* h, hipStream_t st) {
    const int F0 = h->cfg.fc_neurons[0];
    if (h->L == 1 && F0 == 128) return radial_table_check<1, 128>(h, st);
    if (h->L == 2 && F0 == 128) return radial_table_check<2, 128>(h, st);
    if (h->L == 2 && F0 == 192) return radial_table_check<2, 192>(h, st);
    if (h->L == 3 && F0 == 128) return radial_table_check<3, 128>(h, st);
    return DEDF_OK;
}

// one evaluation of the score head on poses already in h->d_Ts (f32) with times in h->d_time
template <int L, int F0, bool EBM = (F0 == kLenEmb)>
int score_impl(dedf_handle* h, int nT, int time_stride, float* ang, float* lin, hipStream_t st) {
    // F0 = 64: no edge time encoding -- the EBM critic (no time at all) or, EBM = false, the score head whose time comes through
    // query_time_encoding alone; 128 / 192: score head with 64 / 128 time channels in the pre-linear
    constexpr bool NOTIME = F0 == kLenEmb;
    static_assert(!EBM || NOTIME, "the EBM head has no time encoding");
    const dedf_config& c = h->cfg;
    const int ns = c.n_scales, nQ = h->nQ;
    const int Nd = nT * nQ;
    constexpr int D = feat_dim<L>();
    const float* nat = h->d_nat.as<float>();
    auto mark = [&]() {
        if (!h->profile) return;
        if (h->ev_used == h->ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; h->ev.push_back(e); }
        (void)hipEventRecord(h->ev[h->ev_used++], st);
    };
    if ((size_t)Nd * D * 4 >= (1ull << 32)) return fail(h, DEDF_ERR_INVALID, "nT*nQ too large for one call (z buffer > 4 GiB); split the pose batch");
    mark();
    // neighbour-search parameters
    NbrParams np{};
    np.key_x = h->d_key_x.as<float>(); np.n_keys = h->n_keys; np.n_scales = ns; np.max_neighbors = c.max_neighbors;
    for (int n = 0; n <= ns; ++n) np.scale_start[n] = h->scale_start[n];
    for (int n = 0; n < ns; ++n) np.r2[n] = c.radii[n] > 0 ? c.radii[n] * c.radii[n] : -1.0f;
    np.qpos = h->d_qpos.as<float>(); np.n_dst = Nd; np.cnt = h->d_cnt.as<int>(); np.off = h->d_off.as<int>(); np.blk = h->d_blk.as<int>();
    np.tile_info = h->d_tile.as<int>(); np.edge_src = h->d_esrc.as<int>(); np.edge_dst = h->d_edst.as<int>(); np.cap = h->edge_cap;
    np.word_start[0] = 0;
    for (int n = 0; n < ns; ++n) np.word_start[n + 1] = np.word_start[n] + (h->scale_start[n + 1] - h->scale_start[n] + 31) / 32;
    np.mask = h->d_mask.as<uint32_t>();
    np.edge_hist = h->profile ? h->d_hist.as<long long>() : nullptr;
    // small batches: pose preparation + word-parallel masks in one launch, single-workgroup count / scan / fill (dedf_misc.h)
    static const int small_max = [] { const char* e = getenv("DEDF_SMALL_BATCH_MAX"); return e ? atoi(e) : kNbrSmallMax; }();      // (experiments)
    bool small = Nd <= small_max && h->small_batch_path;
    for (int n = 0; n < ns; ++n) if (c.radii[n] > 0 && h->scale_start[n + 1] - h->scale_start[n] > c.max_neighbors) small = false;      // the cap could bind
    const double* T64 = h->fused_step ? h->fused_step->T : (const double*)nullptr;
    const int* cnt_used = h->d_cnt.as<int>();
    if (small) {
        const int nblk = (Nd + kNbrBlock - 1) / kNbrBlock;
        const size_t nc = (size_t)ns * Nd, nb = (size_t)ns * nblk;
        if (!h->d_cnt2.ensure(2 * nc * 4) || !h->d_blk2.ensure(2 * nb * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(workspace) failed");
        const int64_t layout = ((int64_t)Nd << 8) | ns;
        if (h->small_layout != layout) {        // first evaluation of this shape: both sets start from zero
            HIPCK(h, hipMemsetAsync(h->d_cnt2.p, 0, h->d_cnt2.bytes, st));
            HIPCK(h, hipMemsetAsync(h->d_blk2.p, 0, h->d_blk2.bytes, st));
            h->small_layout = layout; h->small_parity = 0;
        }
        const int par = h->small_parity;
        np.cnt = h->d_cnt2.as<int>() + par * nc; np.blk = h->d_blk2.as<int>() + par * nb;
        np.zero_cnt = h->d_cnt2.as<int>() + (1 - par) * nc; np.zero_blk = h->d_blk2.as<int>() + (1 - par) * nb;
        h->small_parity = 1 - par;
        cnt_used = np.cnt;
        if constexpr (!EBM) {
            if constexpr (!NOTIME) if (h->tb_step == nullptr) launch_time_bias(h, st, h->d_time.as<float>(), time_stride, time_stride ? nT : 1, h->d_tb.as<float>(), time_stride ? h->d_tile.as<int>() + kFlagTimeVaries : nullptr);
            if (h->tb_step == nullptr && c.query_time_encoding) launch_time_query(h, st, h->d_time.as<float>(), time_stride, time_stride ? nT : 1, h->d_qrows.as<float>());
        }
        mark();
        const dim3 g(nblk, np.word_start[ns] + 1);
        hipLaunchKernelGGL(k_nbr_masks_small<L>, g, dim3(kNbrBlock), 0, st, np, h->d_Ts.as<float>(), T64, h->d_qx.as<float>(), nQ, nT, h->d_pose.as<float>(), h->d_qpos.as<float>());
        // (few destination blocks: the fill pass's bit walk is split over G blocks per destination block, dedf_misc.h)
        const int G = std::max(1, std::min(16, (2 * h->n_cu) / std::max(1, nblk)));
        hipLaunchKernelGGL(k_neighbors<true>, dim3(nblk, G), dim3(kNbrBlock), 0, st, np);
    } else {
    // 1. poses: Wigner-D + transformed query positions
    hipLaunchKernelGGL(k_pose_prep<L>, dim3(nT), dim3(64), 0, st, h->d_Ts.as<float>(), T64, h->d_qx.as<float>(), nQ, h->d_pose.as<float>(), h->d_qpos.as<float>());
    // 2. time embedding -> pre-linear bias rows (EBM head: constant bias rows, uploaded once)
    if constexpr (!EBM) {
        if constexpr (!NOTIME) if (h->tb_step == nullptr) launch_time_bias(h, st, h->d_time.as<float>(), time_stride, time_stride ? nT : 1, h->d_tb.as<float>(), time_stride ? h->d_tile.as<int>() + kFlagTimeVaries : nullptr);
        if (h->tb_step == nullptr && c.query_time_encoding) launch_time_query(h, st, h->d_time.as<float>(), time_stride, time_stride ? nT : 1, h->d_qrows.as<float>());
    }
    mark();
    // 3. neighbour search
    const int nblk = (Nd + kNbrBlock - 1) / kNbrBlock;
    hipLaunchKernelGGL(k_neighbors<false>, dim3(nblk), dim3(kNbrBlock), 0, st, np);
    hipLaunchKernelGGL(k_neighbors<true>, dim3(nblk), dim3(kNbrBlock), 0, st, np);
    }
    mark();
    // 4. fused edge pipeline
    {
        EdgeParams P = edge_params<L, F0>(h, nT, time_stride);
        constexpr int kAll = 1 << 30;
        const bool hp = h->cfg.half_gemm != 0;             // half_gemm (the reference's half_precision knob): single-term fp16 products
        const bool narrow = h->cfg.fc_neurons[1] == 32;    // radial MLP [., 32, 32] (sapien place_*, KeypointExtractor fields) instead of [., 128, 64]
        // Sampler (every pose shares the step's time): the front of the radial network is a function of (scale, length) only -- tabulate
        // it once per launch on a fine length grid with the tile's own code and interpolate per edge (dedf_edge.h: EdgeParams::rtab)
        bool use_tab = false;
        if constexpr (has_radial_table<L, F0>()) {
            // (worth its 34 us generator launch from ~6 rounds of edge tiles on: ~20 edges per destination node -> 8 192 nodes)
            const bool async_tab = h->tab_slot >= 0;      // dedf_sample: this step's table was generated on the side stream (ring slot h->tab_slot)
            use_tab = table_instantiated<L, F0>(h) && h->radial_table != 0 && P.tb_pose_stride == 0 && !h->debug && (Nd >= kRtabMinNodes || h->radial_table == 2 || async_tab);
            if (use_tab) {
                int rc = radial_table_setup(h, P);
                if (rc != DEDF_OK) return rc;
                if (async_tab) {
                    P.rtab = h->d_rtab_ring.as<float>() + (size_t)h->tab_slot * radial_table_rows(h) * 64;
                    P.rtab_out = const_cast<float*>(P.rtab);
                    HIPCK(h, hipStreamWaitEvent(st, h->rt_tab[h->tab_slot], 0));
                }
                if (narrow) {
                    if constexpr (L == 2 && F0 == 128) {
                        if (!async_tab) launch_radial_table<L, F0, 32, 32>(h, P, st, false);
                        launch_edge<L, F0, false, 32, 32, 1>(h, st, P);
                    }
                } else {
                    if (!async_tab) launch_radial_table<L, F0, 128, 64>(h, P, st, false);
                    launch_edge<L, F0, false, 128, 64, 1>(h, st, P);
                }
                if (async_tab) HIPCK(h, hipEventRecord(h->rt_used[h->tab_slot], st));      // the slot may be refilled once this edge kernel is done
            }
        }
        // dedf_score (one time PER pose): when all the times are equal -- the reference's own callers evaluate a batch at ONE diffusion time
        // (score_model_base.py:174-177, warmup) -- the table applies just the same.  The times are on the device, so both forms are enqueued behind
        // a launch gate: k_time_bias left tile_info[kFlagTimeVaries] = 1 if a pose's time differs from pose 0's; the table generator, its accuracy
        // check and the table-reading kernel run when it is 0, the per-pose-time kernel below when it is 1; the other one returns at once.
        if constexpr (has_radial_table<L, F0>()) {
            const bool gated = !use_tab && time_stride != 0 && h->tb_step == nullptr && table_instantiated<L, F0>(h) && h->radial_table != 0 && !h->debug &&
                               (Nd >= kRtabMinNodes || h->radial_table == 2);
            if (gated) {
                EdgeParams Q = P;
                Q.tb_pose_stride = 0; Q.tb_bytes = (uint32_t)((size_t)ns * F0 * 4);      // pose 0's rows stand for all
                Q.gate = h->d_tile.as<int>() + kFlagTimeVaries; Q.gate_want = 0;
                int rc = radial_table_setup(h, Q);
                if (rc != DEDF_OK) return rc;
                HIPCK(h, hipMemsetAsync(h->d_rtab_err.p, 0, kMaxScales * 4, st));
                if (narrow) {
                    if constexpr (L == 2 && F0 == 128) {
                        launch_radial_table<L, F0, 32, 32>(h, Q, st, true);
                        launch_edge<L, F0, false, 32, 32, 1>(h, st, Q);
                    }
                } else {
                    launch_radial_table<L, F0, 128, 64>(h, Q, st, true);
                    launch_edge<L, F0, false, 128, 64, 1>(h, st, Q);
                }
                P.gate = Q.gate; P.gate_want = 1;
            }
        }
        if (use_tab) {
        } else if constexpr (L == 3) {          // lmax 3: full precision, [., 128, 64] (score head, EBM critic) or [64, 32, 32] (context-free fields)
            static_assert(F0 == 128 || F0 == 64, "lmax 3 instantiations");
            if constexpr (F0 == 64) {
                if (narrow) { if (hp) launch_edge<3, 64, true, 32, 32, 0>(h, st, P); else launch_edge<3, 64, false, 32, 32, 0>(h, st, P); }
                else if (hp) launch_edge<3, 64, true, 128, 64, 0>(h, st, P);
                else launch_edge<3, 64, false, 128, 64, 0>(h, st, P);
            } else if (hp) launch_edge<3, 128, true, 128, 64, 0>(h, st, P);
            else launch_edge<3, 128, false, 128, 64, 0>(h, st, P);
        } else if constexpr (F0 == 128) {
            if (narrow) {         // narrow radial MLP (sapien place_*)
                if (hp) launch_edge<L, F0, true, 32, 32, 0>(h, st, P);
                else launch_edge<L, F0, false, 32, 32, 0>(h, st, P);
            } else if (hp) launch_edge<L, F0, true, 128, 64, 0>(h, st, P);
            else launch_edge<L, F0, false, 128, 64, 0>(h, st, P);
        } else if constexpr (F0 == 64 && L == 2) {
            if (narrow) { if (hp) launch_edge<L, F0, true, 32, 32, 0>(h, st, P); else launch_edge<L, F0, false, 32, 32, 0>(h, st, P); }      // KeypointExtractor fields
            else if (hp) launch_edge<L, F0, true, 128, 64, 0>(h, st, P);
            else launch_edge<L, F0, false, 128, 64, 0>(h, st, P);
        } else if (hp) launch_edge<L, F0, true, 128, 64, 0>(h, st, P);
        else launch_edge<L, F0, false, 128, 64, 0>(h, st, P);
    }
    mark();
    // 5. joint softmax + aggregation
    hipLaunchKernelGGL(k_aggregate<L>, dim3((Nd + 3) / 4), dim3(256), 0, st, h->d_eout.as<float>(), cnt_used, h->d_off.as<int>(),
                       h->d_tile.as<int>(), Nd, ns, h->d_z.as<float>());
    mark();
    // 6. node epilogue + score tensor products
    const float* node_spin = nullptr;
    {
        NodeParams P{};
        P.z = h->d_z.as<float>(); P.z_bytes = (uint32_t)((size_t)Nd * D * 4);
        P.qf = h->d_qf.as<float>(); P.qf_bytes = (uint32_t)((size_t)nQ * D * 4);
        P.pose = h->d_pose.as<float>(); P.pose_bytes = (uint32_t)((size_t)nT * pose_rec<L>() * 4);
        P.qx = h->d_qx.as<float>(); P.qw = h->d_qw.as<float>(); P.nQ = nQ; P.n_nodes = Nd; P.lin_mult = c.lin_mult;
        P.W = h->d_node_w.as<float>(); P.W_bytes = (uint32_t)h->d_node_w.bytes;
        const NodeOffsets& o = h->no;
        for (int l = 0; l < 4; ++l) {
            P.o_A_proj[l] = o.o_A_proj[l]; P.o_ln_w[l] = o.o_ln_w[l]; P.o_A_f1[l] = o.o_A_f1[l]; P.o_A_f2[l] = o.o_A_f2[l];
            P.o_A_proj_l[l] = o.o_A_proj_l[l]; P.o_A_f1_l[l] = o.o_A_f1_l[l]; P.o_A_f2_l[l] = o.o_A_f2_l[l];
        }
        P.sc = o.sc;
        P.o_b_proj0 = o.o_b_proj0; P.o_ln_b0 = o.o_ln_b0; P.o_b_f1 = o.o_b_f1; P.o_b_f2 = o.o_b_f2;
        if constexpr (!EBM) for (int t = 0; t < 2; ++t) {
            for (int p = 0; p < 16; ++p) { P.o_A_s[t][p] = o.o_A_s[t][p]; P.o_A_s_l[t][p] = o.o_A_s_l[t][p]; }
            P.o_A_sl[t][0] = o.o_A_sl[t][0]; P.o_A_sl[t][1] = o.o_A_sl[t][1]; P.o_b_sl[t] = o.o_b_sl[t];
            P.o_A_sl_l[t][0] = o.o_A_sl_l[t][0]; P.o_A_sl_l[t][1] = o.o_A_sl_l[t][1];
        }
        if (c.query_time_encoding) { int ps = 0; P.skip1 = query_time_rows(h, time_stride, ps) + mul_of(0); P.skip1_stride = ps; }
        P.node_out = h->d_nout.as<float>();
        if ((h->debug || h->want_field) && h->d_dbge.ensure((size_t)Nd * D * 4) && h->d_dbgf.ensure((size_t)Nd * D * 4)) { P.dbg_emb = h->d_dbge.as<float>(); P.dbg_field = h->d_dbgf.as<float>(); }
        else if (h->want_field) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(field) failed");
        const int ntiles = (Nd + 31) / 32;
        // small batches: two waves per node tile, one score tensor product each (dedf_node.h: NodeParams::split) while both fit the chip at once
        static const bool split_on = [] { const char* e = getenv("DEDF_NODE_SPLIT"); return !(e && atoi(e) == 0); }();
        if constexpr (!EBM) {
            if (split_on && !h->cfg.half_gemm && !h->debug && !h->want_field && 2 * ntiles <= h->n_cu * 4 && h->d_nspin.ensure((size_t)Nd * 16)) {
                P.split = 1; P.node_spin = h->d_nspin.as<float>(); node_spin = P.node_spin;
            }
        }
        if (P.split) {
            if constexpr (!EBM) DEDF_LAUNCH_PERSISTENT((k_node<L, EBM>), 2 * ntiles, st, P);
        } else
        if constexpr (L == 3) {      // (balanced grid measured at lmax 3: 0.477 -> 0.482 ms, not used)
            if (h->cfg.half_gemm) DEDF_LAUNCH_PERSISTENT((k_node<L, EBM, true>), ntiles, st, P);
            else DEDF_LAUNCH_PERSISTENT((k_node<L, EBM>), ntiles, st, P);
        } else if (h->cfg.half_gemm) DEDF_LAUNCH_PERSISTENT((k_node<L, EBM, true>), balanced_blocks(ntiles, h->n_cu * waves_per_cu<(k_node<L, EBM, true>)>()), st, P);
        else DEDF_LAUNCH_PERSISTENT((k_node<L, EBM>), balanced_blocks(ntiles, h->n_cu * waves_per_cu<(k_node<L, EBM>)>()), st, P);
    }
    mark();
    // 7. per-pose reduction
    if constexpr (EBM) hipLaunchKernelGGL(k_energy_reduce, dim3((nT + 127) / 128), dim3(128), 0, st, h->d_nout.as<float>(), nT, nQ, ang, h->d_tile.as<int>());
    else if (h->fused_step) hipLaunchKernelGGL(k_reduce_langevin, dim3(nT), dim3(64), 0, st, h->d_nout.as<float>(), nQ, ang, lin, *h->fused_step, h->d_tile.as<int>(), node_spin);
    else hipLaunchKernelGGL(k_pose_reduce, dim3(nT), dim3(64), 0, st, h->d_nout.as<float>(), nT, nQ, ang, lin, h->d_tile.as<int>(), node_spin);
    mark();
    if (h->profile) { h->prof_evals += 1; h->prof_dst += Nd; }
    if (hipGetLastError() != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, "kernel launch failed");
    h->last_nT = nT;
    h->last_stream = st;
    h->stats_fresh = false;
    return DEDF_OK;
}

int score_dispatch(dedf_handle* h, int nT, int time_stride, float* ang, float* lin, hipStream_t st) {
    const int F0 = h->cfg.fc_neurons[0];
    if (h->L == 1) {
        if (F0 == 64) return h->cfg.ebm ? score_impl<1, 64, true>(h, nT, time_stride, ang, lin, st) : score_impl<1, 64, false>(h, nT, time_stride, ang, lin, st);
        if (F0 == 128) return score_impl<1, 128>(h, nT, time_stride, ang, lin, st);
        return fail(h, DEDF_ERR_UNSUPPORTED, "lmax 1 with a 128-channel time embedding is not instantiated");
    }
    if (h->L == 3) {
        if (F0 == 64) return h->cfg.ebm ? score_impl<3, 64, true>(h, nT, time_stride, ang, lin, st) : score_impl<3, 64, false>(h, nT, time_stride, ang, lin, st);
        if (F0 == 128) return score_impl<3, 128>(h, nT, time_stride, ang, lin, st);
        return fail(h, DEDF_ERR_UNSUPPORTED, "lmax 3 with a 128-channel time embedding is not instantiated");
    }
    if (F0 == 64) return h->cfg.ebm ? score_impl<2, 64, true>(h, nT, time_stride, ang, lin, st) : score_impl<2, 64, false>(h, nT, time_stride, ang, lin, st);
    if (F0 == 128) return score_impl<2, 128>(h, nT, time_stride, ang, lin, st);
    return score_impl<2, 192>(h, nT, time_stride, ang, lin, st);
}

}  // namespace

// ===========================================================================================================================
extern "C" {

const char* dedf_version(void) { return "dedf 0.4.0 (gfx950)"; }
int dedf_abi_version(void) { return DEDF_ABI_VERSION; }
size_t dedf_struct_size(int which) {
    switch (which) {
        case 0: return sizeof(dedf_config);
        case 1: return sizeof(dedf_schedule);
        case 2: return sizeof(dedf_stats);
        case 3: return sizeof(dedf_profile);
        default: return 0;
    }
}

int dedf_param_count(const dedf_config* cfg) {
    std::string why;
    if (check_config(cfg, why) != DEDF_OK) return -1;
    return (int)spec_for(cfg).entries.size();
}
const char* dedf_param_name(const dedf_config* cfg, int i) {
    static thread_local std::string s;
    std::string why;
    if (check_config(cfg, why) != DEDF_OK) return nullptr;
    const ParamSpec S = spec_for(cfg);
    if (i < 0 || i >= (int)S.entries.size()) return nullptr;
    s = S.entries[i].name;
    return s.c_str();
}
size_t dedf_param_numel(const dedf_config* cfg, int i) {
    std::string why;
    if (check_config(cfg, why) != DEDF_OK) return 0;
    const ParamSpec S = spec_for(cfg);
    if (i < 0 || i >= (int)S.entries.size()) return 0;
    return S.entries[i].numel;
}

int dedf_create(const dedf_config* cfg, const float* params, size_t n_params, dedf_handle** out) {
    if (!out) return DEDF_ERR_INVALID;
    *out = nullptr;
    auto h = std::make_unique<dedf_handle>();
    std::string why;
    int rc = check_config(cfg, why);
    if (rc != DEDF_OK) { fprintf(stderr, "dedf_create: %s\n", why.c_str()); return rc; }
    h->cfg = *cfg;
    h->L = cfg->lmax;
    h->host_only = cfg->device < 0;
    if (const char* e = getenv("DEDF_RADIAL_TABLE")) h->radial_table = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("DEDF_RADIAL_TABLE_BOUND")) h->rtab_err_bound = (float)atof(e);
    if (const char* e = getenv("DEDF_SMALL_BATCH")) h->small_batch_path = atoi(e) != 0;
    if (const char* e = getenv("DEDF_RTAB_ASYNC")) h->rtab_async = atoi(e) != 0;
    if (const char* e = getenv("DEDF_RTAB_FIN")) h->rtab_fin = std::max(64, atoi(e));
    if (const char* e = getenv("DEDF_RTAB_INF")) h->rtab_inf = std::max(64, atoi(e));
    h->so2 = so2_instantiated(*cfg);
    if (const char* e = getenv("DEDF_SO2")) h->so2 = h->so2 && (atoi(e) != 0 || !general_instantiated(*cfg));      // (A/B where both forms exist)
    const IrrepsRT T(h->L, cfg->unet_layer != 0), K(h->L, true);
    h->spec = build_spec(T, h->cfg);
    h->kspec = build_spec(K, h->cfg);
    if (!(h->L == 1 ? irreps_consistent<1>(K) : (h->L == 2 ? irreps_consistent<2>(K) : irreps_consistent<3>(K)))) {
        fprintf(stderr, "dedf_create: run-time irreps bookkeeping (dedf_pack.h::IrrepsRT) disagrees with dedf_net.h\n");
        return DEDF_ERR_RUNTIME;
    }
    if (!params || n_params != h->spec.total) {
        fprintf(stderr, "dedf_create: expected %zu parameters, got %zu\n", h->spec.total, n_params);
        return DEDF_ERR_INVALID;
    }
    h->params.assign(params, params + n_params);
    if (cfg->unet_layer && h->L == 3) {
        // The lmax-3 layer kernels skip the lane-local work on the channels p with p % 4 >= q of the 16x3e block (q = 2; 1 for the narrow
        // instantiations, unet_narrow) -- include/dedf.h.  A blob whose true 3e channels sit anywhere else gives silently wrong
        // features: refuse it.  Checked on the tensors whose l = 3 block is addressable by name -- the three 16 x 16 LinearRS blocks and the
        // post-norm's per-channel weights; unet_pad.py::place puts every tensor's channels at the same positions.
        const int q = cfg->unet_narrow ? 1 : 2;      // (what the instantiation that will run skips: dedf_net.h::pad_reg)
        auto live = [&](int p) { return p % 4 < q; };
        const char* bad = nullptr;
        for (const char* nm : {"gnn.linear_src.tp.weight", "gnn.linear_dst.tp.weight", "gnn.ga.proj.tp.weight"}) {
            const float* w = h->spec.get(params, nm) + (64 * 64 + 32 * 32 + 16 * 16);
            for (int u = 0; u < 16 && !bad; ++u) for (int v = 0; v < 16; ++v) if (!(live(u) && live(v)) && w[u * 16 + v] != 0.0f) { bad = nm; break; }
        }
        {
            const float* a = h->spec.get(params, "gnn.norm_2.affine_weight") + (64 + 32 + 16);
            for (int p = 0; p < 16 && !bad; ++p) if (!live(p) && a[p] != 0.0f) bad = "gnn.norm_2.affine_weight";
        }
        if (bad) {
            fprintf(stderr, "dedf_create: UNet layer at lmax 3: %s is non-zero on a channel p with p %% 4 >= %d of the 16x3e block -- the true 3e channels must sit at the "
                            "positions of diffusion_edf_amd/unet_pad.py::place (8x3e: 0, 1, 4, 5, 8, 9, 12, 13; 4x3e: 0, 4, 8, 12), the kernels skip the others\n", bad, q);
            return DEDF_ERR_INVALID;
        }
    }
    try {
        h->kparams = pad_params(h->cfg, T, h->spec, K, h->kspec, h->params.data());
        if (h->L == 1) pack_all<1>(h.get()); else if (h->L == 2) pack_all<2>(h.get()); else pack_all<3>(h.get());
    } catch (const std::exception& e) {
        fprintf(stderr, "dedf_create: %s\n", e.what());
        return DEDF_ERR_INVALID;
    }
    if (!h->no.layout_ok) {
        fprintf(stderr, "dedf_create: packed node image does not match the compile-time layout (dedf_net.h::kNodeLayout)\n");
        return DEDF_ERR_RUNTIME;
    }
    if (!h->host_only) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device) {
            fprintf(stderr, "dedf_create: HIP device %d not available (this library has no CPU path)\n", cfg->device);
            return DEDF_ERR_RUNTIME;
        }
        DeviceGuard guard(cfg->device);
        if (!guard.ok) return DEDF_ERR_RUNTIME;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
            h->n_cu = prop.multiProcessorCount;
            if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
                fprintf(stderr, "dedf_create: device is %s; this library is built for gfx950 only\n", prop.gcnArchName);
                return DEDF_ERR_UNSUPPORTED;
            }
        }
        rc = upload_weights(h.get());
        if (rc != DEDF_OK) { fprintf(stderr, "dedf_create: %s\n", h->err.c_str()); return rc; }
    }
    *out = h.release();
    return DEDF_OK;
}

void dedf_destroy(dedf_handle* h) {
    if (!h) return;
    // workspace links (dedf_layer_share_workspace): a destroyed owner detaches its borrowers (they allocate their own workspace on their next
    // call instead of touching freed memory), a destroyed borrower leaves its owner's list
    if (h->h_flags) { (void)hipHostFree(h->h_flags); (void)hipEventDestroy(h->ev_flags); }
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    for (dedf_handle* b : h->ws_borrowers) b->ws_owner = nullptr;
    if (h->ws_owner) {
        auto& v = h->ws_owner->ws_borrowers;
        v.erase(std::remove(v.begin(), v.end(), h), v.end());
    }
    delete h;
}

const char* dedf_last_error(const dedf_handle* h) { return h ? h->err.c_str() : "null handle"; }

int dedf_set_key_clouds(dedf_handle* h, int n_scales, const int* n_pts, const float* const* x, const float* const* f, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handle: use dedf_layer_forward");
    if (n_scales != h->cfg.n_scales) return fail(h, DEDF_ERR_INVALID, "len(key_pcd_multiscale) != n_scales");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    { const int rcp = check_pending(h, true); if (rcp != DEDF_OK) return rcp; }      // (this entry point synchronises anyway)
    const size_t D = feat_dim_rt(h->L), Dt = true_feat_dim_rt(h->L);      // key features arrive in the true shapes; the message leaves in the kernel layout
    int total = 0;
    for (int n = 0; n < n_scales; ++n) {
        if (n_pts[n] < 0 || (n_pts[n] > 0 && (!x[n] || !f[n]))) return fail(h, DEDF_ERR_INVALID, "bad key cloud");
        h->scale_start[n] = total;
        total += n_pts[n];
    }
    h->scale_start[n_scales] = total;
    if (total <= 0) return fail(h, DEDF_ERR_INVALID, "empty key clouds");
    if ((size_t)total * D * 4 >= (1ull << 32)) return fail(h, DEDF_ERR_INVALID, "key clouds too large");
    h->n_keys = total;
    if (!h->d_key_x.ensure((size_t)total * 3 * 4) || !h->d_key_f.ensure((size_t)total * Dt * 4) || !h->d_msg.ensure((size_t)total * D * 4))
        return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(key clouds) failed");
    for (int n = 0; n < n_scales; ++n) {
        if (n_pts[n] == 0) continue;
        HIPCK(h, hipMemcpyAsync(h->d_key_x.as<float>() + (size_t)h->scale_start[n] * 3, x[n], (size_t)n_pts[n] * 3 * 4, hipMemcpyDeviceToDevice, st));
        HIPCK(h, hipMemcpyAsync(h->d_key_f.as<float>() + (size_t)h->scale_start[n] * Dt, f[n], (size_t)n_pts[n] * Dt * 4, hipMemcpyDeviceToDevice, st));
    }
    const float* nat = h->d_nat.as<float>();
    if (h->L == 1)
        hipLaunchKernelGGL(k_src_message<1>, dim3(total), dim3(64), 0, st, h->d_key_f.as<float>(), total, nat + h->nat_lnw, nat + h->nat_lnb,
                           nat + h->nat_wsrc, nat + h->nat_bsrc, h->d_msg.as<float>(), 0, 0, 0, 0, h->eo.msg_scale);
    else if (h->L == 2)
        hipLaunchKernelGGL(k_src_message<2>, dim3(total), dim3(64), 0, st, h->d_key_f.as<float>(), total, nat + h->nat_lnw, nat + h->nat_lnb,
                           nat + h->nat_wsrc, nat + h->nat_bsrc, h->d_msg.as<float>(), 0, 0, 0, 0, h->eo.msg_scale);
    else
        hipLaunchKernelGGL((k_src_message<3, true, true>), dim3(total), dim3(64), 0, st, h->d_key_f.as<float>(), total, nat + h->nat_lnw, nat + h->nat_lnb,
                           nat + h->nat_wsrc, nat + h->nat_bsrc, h->d_msg.as<float>(), 0, 0, 0, 0, h->eo.msg_scale);
    // edge-workspace sizing: the mean number of same-scale key points within the scale's radius of a key point, summed over the scales
    {
        const int ns = n_scales;
        if (!h->d_deg.ensure((size_t)kMaxScales * (8 + 4 + 4) + 16)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(key clouds) failed");
        unsigned long long* deg = h->d_deg.as<unsigned long long>();
        int* sstart = reinterpret_cast<int*>(deg + kMaxScales);
        float* r2 = reinterpret_cast<float*>(sstart + kMaxScales + 1);
        float hr2[kMaxScales];
        for (int n = 0; n < ns; ++n) hr2[n] = h->cfg.radii[n] > 0 ? h->cfg.radii[n] * h->cfg.radii[n] : -1.0f;
        HIPCK(h, hipMemsetAsync(deg, 0, kMaxScales * 8, st));
        HIPCK(h, hipMemcpyAsync(sstart, h->scale_start, (ns + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(h, hipMemcpyAsync(r2, hr2, ns * sizeof(float), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_self_degree, dim3((total + 255) / 256), dim3(256), 0, st, h->d_key_x.as<float>(), total, ns, sstart, r2, h->cfg.max_neighbors, deg);
        unsigned long long hdeg[kMaxScales];
        HIPCK(h, hipMemcpyAsync(hdeg, deg, kMaxScales * 8, hipMemcpyDeviceToHost, st));
        HIPCK(h, hipStreamSynchronize(st));
        double est = 0.0;
        for (int n = 0; n < ns; ++n) if (n_pts[n] > 0) est += (double)hdeg[n] / (double)n_pts[n];
        h->est_degree = est;
        h->auto_per_dst = std::max<int64_t>(96, (int64_t)std::ceil(1.5 * est));
    }
    h->have_keys = true;
    h->have_key_w = false;
    return DEDF_OK;
}

int dedf_set_key_weights(dedf_handle* h, int n_scales, const int* n_pts, const float* const* w, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->have_keys) return fail(h, DEDF_ERR_INVALID, "dedf_set_key_clouds must be called first");
    if (n_scales != h->cfg.n_scales || !n_pts || !w) return fail(h, DEDF_ERR_INVALID, "n_scales mismatch / null arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    for (int n = 0; n < n_scales; ++n)
        if (n_pts[n] != h->scale_start[n + 1] - h->scale_start[n] || (n_pts[n] > 0 && !w[n])) return fail(h, DEDF_ERR_INVALID, "key weights do not match the key clouds");
    if (!h->d_key_w.ensure((size_t)h->n_keys * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(key weights) failed");
    for (int n = 0; n < n_scales; ++n)
        if (n_pts[n] > 0) HIPCK(h, hipMemcpyAsync(h->d_key_w.as<float>() + h->scale_start[n], w[n], (size_t)n_pts[n] * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipStreamSynchronize(st));
    h->have_key_w = true;
    return DEDF_OK;
}

int dedf_set_query(dedf_handle* h, int nQ, const float* x, const float* f, const float* w, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handle: use dedf_layer_forward");
    if (nQ <= 0 || !x || !f) return fail(h, DEDF_ERR_INVALID, "bad query cloud");
    if (!w) return fail(h, DEDF_ERR_INVALID, "query_pcd.w is required (score_head.py:156-157)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    { const int rcp = check_pending(h, true); if (rcp != DEDF_OK) return rcp; }
    const size_t D = feat_dim_rt(h->L), Dt = true_feat_dim_rt(h->L);
    if (!h->d_qx.ensure((size_t)nQ * 3 * 4) || !h->d_qf.ensure((size_t)nQ * D * 4) || !h->d_qw.ensure((size_t)nQ * 4))
        return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(query) failed");
    HIPCK(h, hipMemcpyAsync(h->d_qx.p, x, (size_t)nQ * 3 * 4, hipMemcpyDeviceToDevice, st));
    if (h->L == 3) {      // true shapes (8x3e) -> kernel layout (16x3e, zero-padded)
        if (!h->d_qf_true.ensure((size_t)nQ * Dt * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(query) failed");
        HIPCK(h, hipMemcpyAsync(h->d_qf_true.p, f, (size_t)nQ * Dt * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_pad_features<3>, dim3((unsigned)(((size_t)nQ * D + 255) / 256)), dim3(256), 0, st, h->d_qf_true.as<float>(), h->d_qf.as<float>(), nQ);
    } else
    HIPCK(h, hipMemcpyAsync(h->d_qf.p, f, (size_t)nQ * D * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemcpyAsync(h->d_qw.p, w, (size_t)nQ * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipStreamSynchronize(st));
    h->nQ = nQ;
    h->have_query = true;
    return DEDF_OK;
}

int dedf_score(dedf_handle* h, int nT, const float* Ts, const float* time, float* ang, float* lin, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handle: use dedf_layer_forward");
    if (!h->have_keys || !h->have_query) return fail(h, DEDF_ERR_INVALID, "set_key_clouds / set_query must be called first");
    if (h->cfg.use_src_point_attn && !h->have_key_w) return fail(h, DEDF_ERR_INVALID, "use_src_point_attn: the key clouds carry no point weights (dedf_set_key_weights; gnn_block.py:191-192)");
    if (nT <= 0 || !Ts || !time || !ang || !lin) return fail(h, DEDF_ERR_INVALID, "bad arguments");
    if (h->cfg.ebm) return fail(h, DEDF_ERR_UNSUPPORTED, "EBM head: the score is the autograd of the energy (score_head_ebm.py:203-217), "
                                                          "which needs a backward pass and is not on the accelerated path; use dedf_energy");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    int rc = check_pending(h, false);
    if (rc != DEDF_OK) return rc;
    rc = ensure_workspace(h, nT);
    if (rc != DEDF_OK) return rc;
    DEDF_CLEAR_FLAGS(h, st);
    HIPCK(h, hipMemcpyAsync(h->d_Ts.p, Ts, (size_t)nT * 7 * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemcpyAsync(h->d_time.p, time, (size_t)nT * 4, hipMemcpyDeviceToDevice, st));
    rc = score_dispatch(h, nT, 1, ang, lin, st);
    return rc != DEDF_OK ? rc : post_flags(h, st);
}

int dedf_energy(dedf_handle* h, int nT, const float* Ts, const float* time, float* energy, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->cfg.ebm) return fail(h, DEDF_ERR_UNSUPPORTED, "dedf_energy needs a handle created with dedf_config.ebm = 1");
    if (!h->have_keys || !h->have_query) return fail(h, DEDF_ERR_INVALID, "set_key_clouds / set_query must be called first");
    if (h->cfg.use_src_point_attn && !h->have_key_w) return fail(h, DEDF_ERR_INVALID, "use_src_point_attn: the key clouds carry no point weights (dedf_set_key_weights; gnn_block.py:191-192)");
    if (nT <= 0 || !Ts || !energy) return fail(h, DEDF_ERR_INVALID, "bad arguments");
    (void)time;   // the critic has no time encoding (configs/*/pick_ebm/score_model_configs.yaml:8-9; agent.py:170)
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    int rc = check_pending(h, false);
    if (rc != DEDF_OK) return rc;
    rc = ensure_workspace(h, nT);
    if (rc != DEDF_OK) return rc;
    DEDF_CLEAR_FLAGS(h, st);
    HIPCK(h, hipMemcpyAsync(h->d_Ts.p, Ts, (size_t)nT * 7 * 4, hipMemcpyDeviceToDevice, st));
    rc = score_dispatch(h, nT, 0, energy, nullptr, st);
    return rc != DEDF_OK ? rc : post_flags(h, st);
}

int dedf_field(dedf_handle* h, int n, const float* x, float* field_out, float* emb_out, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->cfg.ebm || h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "dedf_field needs a handle created with dedf_config.ebm = 1 (a field without context encoding)");
    if (h->cfg.use_src_point_attn) return fail(h, DEDF_ERR_UNSUPPORTED, "dedf_field: use_src_point_attn does not apply");
    if (!h->have_keys) return fail(h, DEDF_ERR_INVALID, "set_key_clouds must be called first");
    if (n <= 0 || !x || !field_out) return fail(h, DEDF_ERR_INVALID, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    const size_t D = feat_dim_rt(h->L), Dt = true_feat_dim_rt(h->L);
    // the points become the query cloud of ONE identity pose; their features / weights only feed the energy, which is discarded
    if (!h->d_qx.ensure((size_t)n * 3 * 4) || !h->d_qf.ensure((size_t)n * D * 4) || !h->d_qw.ensure((size_t)n * 4))
        return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(query) failed");
    HIPCK(h, hipMemcpyAsync(h->d_qx.p, x, (size_t)n * 3 * 4, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemsetAsync(h->d_qf.p, 0, (size_t)n * D * 4, st));
    HIPCK(h, hipMemsetAsync(h->d_qw.p, 0, (size_t)n * 4, st));
    h->nQ = n;
    h->have_query = false;
    int rc = ensure_workspace(h, 1);
    if (rc != DEDF_OK) return rc;
    DEDF_CLEAR_FLAGS(h, st);
    static const float kIdentity[7] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    HIPCK(h, hipMemcpyAsync(h->d_Ts.p, kIdentity, sizeof(kIdentity), hipMemcpyHostToDevice, st));
    h->want_field = true;
    rc = score_dispatch(h, 1, 0, h->d_ang.as<float>(), nullptr, st);
    h->want_field = false;
    if (rc != DEDF_OK) return rc;
    const int nb = (int)(((size_t)n * Dt + 255) / 256);
    if (h->L == 3) {
        hipLaunchKernelGGL(k_internal_to_ref<3>, dim3(nb), dim3(256), 0, st, h->d_dbgf.as<float>(), field_out, n);
        if (emb_out) hipLaunchKernelGGL(k_internal_to_ref<3>, dim3(nb), dim3(256), 0, st, h->d_dbge.as<float>(), emb_out, n);
    } else if (h->L == 1) {
        hipLaunchKernelGGL(k_internal_to_ref<1>, dim3(nb), dim3(256), 0, st, h->d_dbgf.as<float>(), field_out, n);
        if (emb_out) hipLaunchKernelGGL(k_internal_to_ref<1>, dim3(nb), dim3(256), 0, st, h->d_dbge.as<float>(), emb_out, n);
    } else {
        hipLaunchKernelGGL(k_internal_to_ref<2>, dim3(nb), dim3(256), 0, st, h->d_dbgf.as<float>(), field_out, n);
        if (emb_out) hipLaunchKernelGGL(k_internal_to_ref<2>, dim3(nb), dim3(256), 0, st, h->d_dbge.as<float>(), emb_out, n);
    }
    if (hipGetLastError() != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, "kernel launch failed");
    return DEDF_OK;
}

int dedf_keypoint_weight(const float* field, const float* emb, int n, int stride, const float* skip_W, const float* skip_b, const float* ln_w,
                         const float* ln_b, const float* lin_w, float lin_b, int sigmoid, float mult, float* out, void* stream) {
    if (!field || !emb || n <= 0 || stride < 64 || !skip_W || !skip_b || !ln_w || !ln_b || !lin_w || !out) return DEDF_ERR_INVALID;
    hipLaunchKernelGGL(k_keypoint_weight, dim3((n + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), field, emb, n, stride, skip_W, skip_b,
                       ln_w, ln_b, lin_w, lin_b, sigmoid, mult, out);
    return hipGetLastError() == hipSuccess ? DEDF_OK : DEDF_ERR_RUNTIME;
}

static int sample_once(dedf_handle* h, int nT, const double* T_seed, const dedf_schedule* sched, uint64_t seed, int64_t first_pose_index,
                       const double* noise, double* Ts_out, void* stream, bool* overflowed) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->have_keys || !h->have_query) return fail(h, DEDF_ERR_INVALID, "set_key_clouds / set_query must be called first");
    if (h->cfg.use_src_point_attn && !h->have_key_w) return fail(h, DEDF_ERR_INVALID, "use_src_point_attn: the key clouds carry no point weights (dedf_set_key_weights; gnn_block.py:191-192)");
    if (nT <= 0 || !T_seed || !sched || sched->n_steps < 0 || !Ts_out) return fail(h, DEDF_ERR_INVALID, "bad arguments");
    if (h->cfg.ebm) return fail(h, DEDF_ERR_UNSUPPORTED, "EBM head: sampling needs the energy gradient (backward pass); use dedf_energy");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    int rc = ensure_workspace(h, nT);
    if (rc != DEDF_OK) return rc;
    h->stats_fresh = false;
    if (h->h_pin_floats < (size_t)128 + std::max(sched->n_steps, 1)) {
        if (h->h_pin) (void)hipHostFree(h->h_pin);
        h->h_pin = nullptr;
        h->h_pin_floats = (size_t)128 + std::max(sched->n_steps, 1) + 256;
        HIPCK(h, hipHostMalloc((void**)&h->h_pin, h->h_pin_floats * 4, hipHostMallocDefault));
    }
    const size_t row = (size_t)nT * 7;
    DEDF_CLEAR_FLAGS(h, st);
    HIPCK(h, hipMemcpyAsync(h->d_T64.p, T_seed, row * 8, hipMemcpyDeviceToDevice, st));
    HIPCK(h, hipMemcpyAsync(Ts_out, T_seed, row * 8, hipMemcpyDeviceToDevice, st));
    // every pose shares the step's time: the time-bias rows of all steps come from ONE launch before the loop
    const size_t tb_row = (size_t)h->cfg.n_scales * h->cfg.fc_neurons[0];
    bool use_async = false;
    if (sched->n_steps > 0) {
        if (!h->d_tb_steps.ensure((size_t)sched->n_steps * tb_row * 4) || !h->d_time.ensure((size_t)std::max(nT, sched->n_steps) * 4))
            return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(time rows) failed");
        // (the score head sees the step's time as FLOAT32, like the reference's: score_model_base.py:177 `time = t.repeat(len(T)).type(dtype)`;
        //  the Langevin update below uses the float64 value)
        float* pin_t = reinterpret_cast<float*>(h->h_pin) + 128;
        for (int s = 0; s < sched->n_steps; ++s) pin_t[s] = (float)sched->t[s];
        HIPCK(h, hipMemcpyAsync(h->d_time.p, pin_t, (size_t)sched->n_steps * 4, hipMemcpyHostToDevice, st));
        // (without edge time encoding the rows are constants: h->tb_step then only names the step for query_time_rows)
        if (h->cfg.fc_neurons[0] != kLenEmb) launch_time_bias(h, st, h->d_time.as<float>(), 1, sched->n_steps, h->d_tb_steps.as<float>());
        if (h->cfg.query_time_encoding) {
            if (!h->d_qrows_steps.ensure((size_t)sched->n_steps * kQueryTimeRow * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(time rows) failed");
            launch_time_query(h, st, h->d_time.as<float>(), 1, sched->n_steps, h->d_qrows_steps.as<float>());
        }
        // Accuracy guard of the radial table, once per call: the table of the first, the middle and the last step (the time only shifts the
        // pre-linear's bias rows; what decides the interpolation error is the length encoder) is checked at EVERY interval midpoint of every
        // scale against the exact evaluation; a scale whose largest deviation exceeds the bound evaluates its front per edge in this call.
        const int64_t Nd_call = (int64_t)nT * h->nQ;
        use_async = h->rtab_async && h->radial_table != 0 && !h->debug && radial_table_instantiated_rt(h) && (Nd_call >= kRtabAsyncMinNodes || h->radial_table == 2);
        if (h->radial_table != 0 && (Nd_call >= kRtabMinNodes || h->radial_table == 2 || use_async) && !h->debug) {
            if (h->d_rtab_err.p) HIPCK(h, hipMemsetAsync(h->d_rtab_err.p, 0, kMaxScales * 4, st));
            int last = -1;
            for (int s : {0, sched->n_steps / 2, sched->n_steps - 1}) {
                if (s == last) continue;
                last = s;
                h->tb_step = h->d_tb_steps.as<float>() + (size_t)s * tb_row;
                rc = radial_table_check_dispatch(h, st);
                h->tb_step = nullptr;
                if (rc != DEDF_OK) return rc;
            }
        }
    }
    constexpr int R = dedf_handle::kRtabRing;
    auto gen_ahead = [&](int step, bool wait_used) -> int {      // side stream: the table of `step` into slot step % R
        const int slot = step % R;
        if (wait_used) HIPCK(h, hipStreamWaitEvent(h->side, h->rt_used[slot], 0));
        h->tb_step = h->d_tb_steps.as<float>() + (size_t)step * tb_row;
        const int rc2 = radial_table_async_dispatch(h, slot);
        h->tb_step = nullptr;
        if (rc2 != DEDF_OK) return rc2;
        HIPCK(h, hipEventRecord(h->rt_tab[slot], h->side));
        return DEDF_OK;
    };
    if (use_async) {
        if (!h->side) {
            HIPCK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            for (int i = 0; i < R; ++i) { HIPCK(h, hipEventCreateWithFlags(&h->rt_tab[i], hipEventDisableTiming)); HIPCK(h, hipEventCreateWithFlags(&h->rt_used[i], hipEventDisableTiming)); }
            HIPCK(h, hipEventCreateWithFlags(&h->rt_ready, hipEventDisableTiming));
        }
        if (!h->d_rtab_ring.ensure((size_t)R * radial_table_rows(h) * 64 * 4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(radial table ring) failed");
        // the side stream starts behind the time-bias rows and the guard's launches of the main stream
        HIPCK(h, hipEventRecord(h->rt_ready, st));
        HIPCK(h, hipStreamWaitEvent(h->side, h->rt_ready, 0));
        for (int s = 0; s < std::min(R, sched->n_steps); ++s) { rc = gen_ahead(s, false); if (rc != DEDF_OK) return rc; }
    }
    for (int s = 0; s < sched->n_steps; ++s) {
        // one step = pose prep (reads the f64 state), neighbour count + fill, edge, aggregate, node, reduce + Langevin update
        LangevinParams lp{};
        lp.T = h->d_T64.as<double>();
        lp.t = sched->t[s]; lp.alpha_ang = sched->alpha_ang[s]; lp.alpha_lin = sched->alpha_lin[s]; lp.temperature = sched->temperature[s];
        lp.ang_mult = h->cfg.ang_mult; lp.lin_mult = h->cfg.lin_mult;
        lp.noise = noise ? noise + (size_t)s * 2 * nT * 3 : nullptr;
        lp.seed = seed; lp.first_pose = first_pose_index; lp.step = s;
        lp.traj_out = Ts_out + (size_t)(s + 1) * row; lp.nT = nT;
        h->tb_step = h->d_tb_steps.as<float>() + (size_t)s * tb_row;
        h->fused_step = &lp;
        h->tab_slot = use_async ? s % R : -1;
        rc = score_dispatch(h, nT, 0, h->d_ang.as<float>(), h->d_lin.as<float>(), st);
        h->tb_step = nullptr; h->fused_step = nullptr; h->tab_slot = -1;
        if (rc != DEDF_OK) return rc;
        if (use_async && s + R < sched->n_steps) { rc = gen_ahead(s + R, true); if (rc != DEDF_OK) return rc; }
    }
    HIPCK(h, hipMemcpyAsync(Ts_out + (size_t)(sched->n_steps + 1) * row, h->d_T64.p, row * 8, hipMemcpyDeviceToDevice, st));
    // sticky status words of the whole call: an overflow or a non-finite score in ANY step is reported (the trajectory then holds NaNs
    // from that step on, never stale scores).  They travel to pinned memory in front of the synchronisation.
    memset(h->h_pin, 0, 72 * 4);
    if (sched->n_steps > 0) {
        HIPCK(h, hipMemcpyAsync(h->h_pin, h->d_tile.p, 64 * 4, hipMemcpyDeviceToHost, st));
        if (h->d_rtab_err.p) HIPCK(h, hipMemcpyAsync(h->h_pin + 64, h->d_rtab_err.p, kMaxScales * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCK(h, hipStreamSynchronize(st));
    h->stats_fresh = sched->n_steps > 0;
    const int flags[2] = {h->h_pin[kFlagOverflow], h->h_pin[kFlagNonFinite]};
    h->clear_seen = true;      // (the sampler consumes its own verdicts -- an overflow it repeats with more room is not news for the next dedf_score)
    if (flags[0]) { *overflowed = true; return fail(h, DEDF_ERR_RUNTIME, "edge workspace overflow: raise dedf_config.max_edges"); }
    if (flags[1]) return fail(h, DEDF_ERR_RUNTIME, "non-finite score: an activation left the fp16 operand range of the split-fp16 GEMMs "
                                                   "(or the inputs / poses were not finite)");
    return DEDF_OK;
}

// The automatic edge workspace (dedf_config.max_edges = 0) is sized from the scene (dedf_set_key_clouds: 1.5 x the degree of a query point on the
// scene surface, at least 96 edges per destination node).  A call whose poses crowd a denser spot than that overflows in some step; the call
// is then repeated from its (untouched) seed poses with twice the room -- same seed, same noise, same result as a call that had the room from
// the start.  An explicit max_edges is never overridden.
int dedf_sample(dedf_handle* h, int nT, const double* T_seed, const dedf_schedule* sched, uint64_t seed, int64_t first_pose_index,
                const double* noise, double* Ts_out, void* stream) {
    if (h && !h->host_only && h->flags_pending) {
        DeviceGuard dev_guard__(h->cfg.device);
        const int rcp = check_pending(h, true);
        if (rcp != DEDF_OK) return rcp;
    }
    const int64_t per_dst_before = h ? h->auto_per_dst : 0;
    for (int attempt = 0;; ++attempt) {
        bool overflowed = false;
        const int rc = sample_once(h, nT, T_seed, sched, seed, first_pose_index, noise, Ts_out, stream, &overflowed);
        if (h) h->sample_retries = attempt;
        if (!overflowed || h->cfg.max_edges > 0 || attempt >= 3) {
            if (rc == DEDF_OK) h->err.clear();                                   // (a repeat that succeeded leaves no stale "overflow" text behind)
            else if (overflowed && h->cfg.max_edges <= 0) h->auto_per_dst = per_dst_before;      // every repeat failed: the capacity is not left 8x larger
            return rc;
        }
        h->auto_per_dst *= 2;
    }
}

}  // extern "C"

namespace {
template <int L>
int layer_forward_impl(dedf_handle* h, int n_src, const float* x_src, const float* f_src, int n_dst, const float* x_dst, const float* f_dst,
                       int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst, float* out, void* stream) {
    constexpr size_t D = feat_dim<L>(), REC = edge_rec<L>();
    if ((size_t)n_src * D * 4 >= (1ull << 32) || (size_t)n_dst * D * 4 >= (1ull << 32) || n_edges >= 0x7fffffff - 64)
        return fail(h, DEDF_ERR_INVALID, "graph too large for one call");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    dedf_handle* const w = h->ws_owner ? h->ws_owner : h;      // whose workspace this call runs in (dedf_layer_share_workspace)
    const size_t E = (size_t)std::max<int64_t>(n_edges, 1);
    bool ok = w->d_msg.ensure((size_t)n_src * D * 4) && w->d_msg_dst.ensure((size_t)n_dst * D * 4) && w->d_tile.ensure(64 * 4) &&
              w->d_esrc.ensure(E * 4) && w->d_edst.ensure(E * 4) && w->d_cnt.ensure((size_t)n_dst * 4) && w->d_off.ensure((size_t)n_dst * 4) &&
              w->d_eout.ensure(E * REC * 4) && w->d_z.ensure((size_t)n_dst * D * 4);
    if (!ok) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(workspace) failed");
    HIPCK(h, hipMemsetAsync(w->d_tile.p, 0, 64 * 4, st));
    const float* nat = h->d_nat.as<float>();
    // messages: linear_src(f_src), linear_dst(f_dst) -- no LayerNorm in front of them (block.py:149-153)
    hipLaunchKernelGGL((k_src_message<L, false>), dim3(n_src), dim3(64), 0, st, f_src, n_src, (const float*)nullptr, (const float*)nullptr,
                       nat + h->nat_wsrc, (const float*)nullptr, w->d_msg.as<float>());
    hipLaunchKernelGGL((k_src_message<L, false>), dim3(n_dst), dim3(64), 0, st, f_dst, n_dst, (const float*)nullptr, (const float*)nullptr,
                       nat + h->nat_wdst, nat + h->nat_bdst, w->d_msg_dst.as<float>());
    {
        const int64_t work = std::max<int64_t>(n_edges, n_dst);
        hipLaunchKernelGGL(k_edge_lists, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, edge_src, edge_dst, n_edges, n_src, n_dst,
                           w->d_esrc.as<int>(), w->d_edst.as<int>(), w->d_cnt.as<int>(), w->d_off.as<int>(), w->d_tile.as<int>());
    }
    const dedf_config& c = h->cfg;
    {
        EdgeParams P{};
        P.key_x = x_src; P.qpos = x_dst; P.edge_src = w->d_esrc.as<int>(); P.edge_dst = w->d_edst.as<int>();
        P.tile_info = w->d_tile.as<int>();
        P.msg = w->d_msg.as<float>(); P.msg_bytes = (uint32_t)((size_t)n_src * D * 4);
        P.msg_dst = w->d_msg_dst.as<float>(); P.msg_dst_bytes = (uint32_t)((size_t)n_dst * D * 4);
        P.tb = nullptr; P.tb_bytes = 0; P.tb_pose_stride = 0;
        P.nQ = 1; P.n_scales = 1;
        for (int i = 0; i < 2; ++i) {      // masked LayerNorms of the radial MLP (true widths of a padded model)
            const int full = c.fc_neurons[1 + i], v = c.unet_fc_valid[1 + i] > 0 ? c.unet_fc_valid[1 + i] : full;
            P.ln_inv_n[i] = 1.0f / (float)v; P.ln_pad[i] = (float)(full - v);
        }
        // GaussianRadialBasisLayerFiniteCutoff(cutoff = 0.99 r): offset = 0.01 cutoff; t = (len - offset) / (cutoff - offset)
        const float cutoff = (float)(0.99 * (double)c.radii[0]), offset = (float)(0.01 * (double)cutoff);
        P.radius[0] = cutoff - offset; P.cut_begin[0] = offset; P.cut_div[0] = 1.0f;
        P.ns_lo = 0.0f; P.ns_div = 1.0f; P.len_enc_max_r = 1.0f;
        P.W = h->d_edge_w.as<float>(); P.W_bytes = (uint32_t)h->d_edge_w.bytes;
        const EdgeOffsets& o = h->eo;
        P.o_enc = o.o_enc; P.o_A_pre = o.o_A_pre; P.o_A_pre_l = o.o_A_pre_l; P.o_A_r1_l = o.o_A_r1_l; P.o_A_r2_l = o.o_A_r2_l; P.o_A_r3_l = o.o_A_r3_l; P.o_A_r1 = o.o_A_r1; P.o_b_r1 = o.o_b_r1; P.o_g_r1 = o.o_g_r1; P.o_be_r1 = o.o_be_r1;
        P.o_A_r2 = o.o_A_r2; P.o_b_r2 = o.o_b_r2; P.o_g_r2 = o.o_g_r2; P.o_be_r2 = o.o_be_r2; P.o_A_r3 = o.o_A_r3; P.o_off_r3 = o.o_off_r3;
        P.o_S_lin = o.o_S_lin; P.o_S_val = o.o_S_val;
        P.w_unscale = o.w_unscale; P.u_scale = o.u_scale;
        for (int l = 0; l < 4; ++l) { P.c_lin[l] = o.c_lin[l]; P.c_val[l] = o.c_val[l]; }
        P.o_b_r0 = o.o_b_r0; P.o_b_val0 = o.o_b_val0; P.o_alpha_dot = o.o_alpha_dot;
        P.key_w = nullptr;
        P.out = w->d_eout.as<float>();
        P.dbg_w = nullptr; P.dbg_out = nullptr; P.phase_prof = nullptr;
        static const bool nw_on = [] { const char* e = getenv("DEDF_UNET_NARROW"); return !(e && atoi(e) == 0); }();      // DEDF_UNET_NARROW=0: the general kernels (A/B)
        if (h->cfg.unet_narrow && nw_on) {      // narrow level: the instantiations that skip the structurally zero channels (dedf_net.h::pad_live)
            if (h->cfg.half_gemm) DEDF_LAUNCH_PERSISTENT((k_edge<L, 64, true, 32, 32, true, 0, true, true>), 1 << 30, st, P);
            else DEDF_LAUNCH_PERSISTENT((k_edge<L, 64, false, 32, 32, true, 0, true, true>), 1 << 30, st, P);      // full precision: edge-aligned frame (round 5)
        } else if (h->cfg.half_gemm) DEDF_LAUNCH_PERSISTENT((k_edge<L, 64, true, 32, 32, true, 0, false, true>), 1 << 30, st, P);
        else DEDF_LAUNCH_PERSISTENT((k_edge<L, 64, false, 32, 32, true, 0, false, true>), 1 << 30, st, P);
    }
    hipLaunchKernelGGL(k_aggregate<L>, dim3((n_dst + 3) / 4), dim3(256), 0, st, w->d_eout.as<float>(), w->d_cnt.as<int>(), w->d_off.as<int>(),
                       w->d_tile.as<int>(), n_dst, 1, w->d_z.as<float>());
    {
        NodeParams P{};
        P.z = w->d_z.as<float>(); P.z_bytes = (uint32_t)((size_t)n_dst * D * 4);
        P.f_dst = f_dst; P.f_dst_bytes = (uint32_t)((size_t)n_dst * D * 4);
        P.feat_out = out;
        P.nQ = 1; P.n_nodes = n_dst; P.lin_mult = 1.0f;
        for (int l = 0; l <= L; ++l) P.ln_inv_n[l] = 1.0f / (float)(c.unet_valid[l] > 0 ? c.unet_valid[l] : c.mul[l]);
        P.ln_pad0 = (float)(c.mul[0] - (c.unet_valid[0] > 0 ? c.unet_valid[0] : c.mul[0]));
        P.W = h->d_node_w.as<float>(); P.W_bytes = (uint32_t)h->d_node_w.bytes;
        const NodeOffsets& o = h->no;
        for (int l = 0; l < 4; ++l) {
            P.o_A_proj[l] = o.o_A_proj[l]; P.o_ln_w[l] = o.o_ln_w[l]; P.o_A_f1[l] = o.o_A_f1[l]; P.o_A_f2[l] = o.o_A_f2[l];
            P.o_A_proj_l[l] = o.o_A_proj_l[l]; P.o_A_f1_l[l] = o.o_A_f1_l[l]; P.o_A_f2_l[l] = o.o_A_f2_l[l];
        }
        P.sc = o.sc;
        P.o_b_proj0 = o.o_b_proj0; P.o_ln_b0 = o.o_ln_b0; P.o_b_f1 = o.o_b_f1; P.o_b_f2 = o.o_b_f2;
        const int ntiles = (n_dst + 31) / 32;
        if (h->cfg.half_gemm) DEDF_LAUNCH_PERSISTENT((k_node<L, false, true, true>), balanced_blocks(ntiles, h->n_cu * waves_per_cu<(k_node<L, false, true, true>)>()), st, P);
        else DEDF_LAUNCH_PERSISTENT((k_node<L, false, false, true>), balanced_blocks(ntiles, h->n_cu * waves_per_cu<(k_node<L, false, false, true>)>()), st, P);
    }
    if (h->defer_check) {          // chains of layers: the verdict of the edge-list check is kept on the device for dedf_layer_check
        if (!w->d_sticky.p) {
            if (!w->d_sticky.ensure(4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(flag) failed");
            HIPCK(h, hipMemsetAsync(w->d_sticky.p, 0, 4, st));
        }
        hipLaunchKernelGGL(k_or_flag, dim3(1), dim3(1), 0, st, w->d_tile.as<int>() + kFlagBadEdges, w->d_sticky.as<int>());
        if (hipGetLastError() != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, "kernel launch failed");
        return DEDF_OK;
    }
    if (hipGetLastError() != hipSuccess) return fail(h, DEDF_ERR_RUNTIME, "kernel launch failed");
    HIPCK(h, hipStreamSynchronize(st));
    int bad = 0;
    HIPCK(h, hipMemcpy(&bad, w->d_tile.as<int>() + kFlagBadEdges, 4, hipMemcpyDeviceToHost));
    if (bad) return fail(h, DEDF_ERR_INVALID, "edge lists: edge_dst must be sorted ascending and every index inside its cloud");
    return DEDF_OK;
}
}  // namespace

extern "C" {

int dedf_layer_forward(dedf_handle* h, int n_src, const float* x_src, const float* f_src, int n_dst, const float* x_dst, const float* f_dst,
                       int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst, float* out, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "dedf_layer_forward needs a handle created with dedf_config.unet_layer = 1");
    if (n_src <= 0 || n_dst <= 0 || n_edges < 0 || !x_src || !f_src || !x_dst || !f_dst || !out || (n_edges > 0 && (!edge_src || !edge_dst)))
        return fail(h, DEDF_ERR_INVALID, "bad arguments");
    if (h->L == 3) return layer_forward_impl<3>(h, n_src, x_src, f_src, n_dst, x_dst, f_dst, n_edges, edge_src, edge_dst, out, stream);
    return layer_forward_impl<2>(h, n_src, x_src, f_src, n_dst, x_dst, f_dst, n_edges, edge_src, edge_dst, out, stream);
}

int dedf_set_radial_table(dedf_handle* h, int on) {
    if (!h) return DEDF_ERR_INVALID;
    if (on < 0 || on > 2) return fail(h, DEDF_ERR_INVALID, "dedf_set_radial_table: 0 off, 1 automatic, 2 always");
    h->radial_table = on;
    return DEDF_OK;
}

int dedf_layer_defer_check(dedf_handle* h, int on) {
    if (!h) return DEDF_ERR_INVALID;
    if (!h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handles only");
    h->defer_check = on != 0;
    return DEDF_OK;
}

int dedf_layer_share_workspace(dedf_handle* h, dedf_handle* owner) {
    if (!h) return DEDF_ERR_INVALID;
    if (!h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handles only");
    auto unlink = [&]() {
        if (!h->ws_owner) return;
        auto& v = h->ws_owner->ws_borrowers;
        v.erase(std::remove(v.begin(), v.end(), h), v.end());
        h->ws_owner = nullptr;
    };
    if (owner == nullptr || owner == h) { unlink(); return DEDF_OK; }
    if (!owner->cfg.unet_layer || owner->host_only || h->host_only) return fail(h, DEDF_ERR_INVALID, "dedf_layer_share_workspace: the owner must be a UNet-layer handle with a device");
    if (owner->cfg.device != h->cfg.device) return fail(h, DEDF_ERR_INVALID, "dedf_layer_share_workspace: both handles must live on one device");
    if (owner->ws_owner) return fail(h, DEDF_ERR_INVALID, "dedf_layer_share_workspace: the owner itself borrows a workspace");
    if (!h->ws_borrowers.empty()) return fail(h, DEDF_ERR_INVALID, "dedf_layer_share_workspace: this handle lends its own workspace to others");
    unlink();
    // a verdict this layer still holds from deferred calls of its own moves to the owner's word (one dedf_layer_check collects all)
    if (h->d_sticky.p) {
        DEDF_ON_DEVICE(h);
        if (!owner->d_sticky.p) {
            if (!owner->d_sticky.ensure(4)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc(flag) failed");
            HIPCK(h, hipMemset(owner->d_sticky.p, 0, 4));
        }
        HIPCK(h, hipDeviceSynchronize());
        hipLaunchKernelGGL(k_or_flag, dim3(1), dim3(1), 0, nullptr, h->d_sticky.as<int>(), owner->d_sticky.as<int>());
        HIPCK(h, hipMemsetAsync(h->d_sticky.p, 0, 4, nullptr));
        HIPCK(h, hipDeviceSynchronize());
    }
    h->ws_owner = owner;
    owner->ws_borrowers.push_back(h);
    // this layer's own per-call buffers are no longer needed
    for (DevBuf* b : {&h->d_msg, &h->d_msg_dst, &h->d_esrc, &h->d_edst, &h->d_cnt, &h->d_off, &h->d_eout, &h->d_z}) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr; b->bytes = 0;
    }
    return DEDF_OK;
}

int dedf_layer_check(dedf_handle* h, void* stream) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    if (!h->cfg.unet_layer) return fail(h, DEDF_ERR_UNSUPPORTED, "UNet-layer handles only");
    dedf_handle* const w = h->ws_owner ? h->ws_owner : h;      // layers that share a workspace share the verdict word: one check covers them all
    if (!w->d_sticky.p) return DEDF_OK;                 // nothing has run in deferred mode
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEDF_ON_DEVICE(h);
    int bad = 0;
    HIPCK(h, hipMemcpyAsync(&bad, w->d_sticky.p, 4, hipMemcpyDeviceToHost, st));
    HIPCK(h, hipStreamSynchronize(st));
    if (bad) {
        HIPCK(h, hipMemsetAsync(w->d_sticky.p, 0, 4, st));
        return fail(h, DEDF_ERR_INVALID, "edge lists: edge_dst must be sorted ascending and every index inside its cloud");
    }
    return DEDF_OK;
}

int dedf_get_stats(dedf_handle* h, dedf_stats* out) {
    if (!h || !out) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    memset(out, 0, sizeof(*out));
    h->flags_pending = false;      // the caller reads the verdict himself
    h->clear_seen = true;
    if (h->last_nT == 0) return DEDF_OK;
    int ti[64];
    const bool fresh = h->stats_fresh && h->h_pin;      // right behind a dedf_sample: its status block is on the host already
    if (fresh) memcpy(ti, h->h_pin, sizeof(ti));
    else {
        HIPCK(h, hipStreamSynchronize(h->last_stream));
        HIPCK(h, hipMemcpy(ti, h->d_tile.p, sizeof(ti), hipMemcpyDeviceToHost));
    }
    out->n_dst = (int64_t)h->last_nT * h->nQ;
    for (int n = 0; n < h->cfg.n_scales; ++n) { out->n_edges[n] = ti[kEdgeCountWord + n]; out->n_edges_total += out->n_edges[n]; }
    out->overflow = ti[40] | ti[kFlagOverflow];
    out->nonfinite = ti[kFlagNonFinite];
    out->sample_retries = h->sample_retries;
    out->edges_per_dst_capacity = h->cfg.max_edges > 0 ? 0 : (int)h->auto_per_dst;
    if (h->d_rtab_err.p) {
        if (fresh) memcpy(out->rtab_err, h->h_pin + 64, kMaxScales * 4);
        else HIPCK(h, hipMemcpy(out->rtab_err, h->d_rtab_err.p, kMaxScales * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < h->cfg.n_scales; ++n) if (!(out->rtab_err[n] <= radial_table_bound(h, n))) out->rtab_fallback |= 1 << n;
    }
    return DEDF_OK;
}

int dedf_profile_enable(dedf_handle* h, int on) {
    if (!h) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    DEDF_ON_DEVICE(h);
    if (on && !h->d_hist.p) {
        if (!h->d_hist.ensure(8)) return fail(h, DEDF_ERR_RUNTIME, "hipMalloc failed");
        HIPCK(h, hipMemset(h->d_hist.p, 0, 8));
    }
    h->profile = on != 0;
    return DEDF_OK;
}

int dedf_profile_read(dedf_handle* h, dedf_profile* out) {
    if (!h || !out) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    memset(out, 0, sizeof(*out));
    HIPCK(h, hipDeviceSynchronize());
    for (size_t i = 0; i + 6 < h->ev_used; i += 7)
        for (int c = 0; c < DEDF_PROF_CLASSES; ++c) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, h->ev[i + c], h->ev[i + c + 1]) == hipSuccess) out->ms[c] += ms;
        }
    out->n_evals = h->prof_evals;
    out->n_dst = h->prof_dst;
    if (h->d_hist.p) {
        long long e = 0;
        HIPCK(h, hipMemcpy(&e, h->d_hist.p, 8, hipMemcpyDeviceToHost));
        out->n_edges = e;
        HIPCK(h, hipMemset(h->d_hist.p, 0, 8));
    }
    h->ev_used = 0; h->prof_evals = 0; h->prof_dst = 0;
    return DEDF_OK;
}

int dedf_debug_enable(dedf_handle* h, int on) {
    if (!h) return DEDF_ERR_INVALID;
    h->debug = on != 0;
    return DEDF_OK;
}

int dedf_debug_copy(dedf_handle* h, const char* name, void* host_dst, size_t max_bytes, size_t* actual_bytes) {
    if (!h || !name) return DEDF_ERR_INVALID;
    if (h->host_only) return fail(h, DEDF_ERR_RUNTIME, "host-only handle");
    const std::string nm(name);
    const size_t D = feat_dim_rt(h->L), REC = edge_rec_rt(h->L), WN = dtp_wn_rt(h->L);      // kernel layout (lmax 3: 16x3e)
    const size_t Nd = (size_t)h->last_nT * h->nQ;
    HIPCK(h, hipDeviceSynchronize());
    int ti[64] = {0};
    if (h->d_tile.p) HIPCK(h, hipMemcpy(ti, h->d_tile.p, sizeof(ti), hipMemcpyDeviceToHost));
    const size_t E = (size_t)ti[16 + h->cfg.n_scales];
    const void* src = nullptr;
    size_t n = 0;
    if (nm == "msg") { src = h->d_msg.p; n = (size_t)h->n_keys * D * 4; }
    else if (nm == "qpos") { src = h->d_qpos.p; n = Nd * 3 * 4; }
    else if (nm == "pose") { src = h->d_pose.p; n = (size_t)h->last_nT * pose_rec_rt(h->L) * 4; }
    else if (nm == "tb") { src = h->d_tb.p; n = (size_t)h->last_nT * h->cfg.n_scales * h->cfg.fc_neurons[0] * 4; }
    else if (nm == "edge_src") { src = h->d_esrc.p; n = E * 4; }
    else if (nm == "edge_dst") { src = h->d_edst.p; n = E * 4; }
    else if (nm == "edge_out") { src = h->d_dbgo.p; n = h->d_dbgo.p ? E * REC * 4 : 0; }        // per-edge records (debug mode)
    else if (nm == "segment_out") { src = h->d_eout.p; n = E * REC * 4; }
    else if (nm == "z") { src = h->d_z.p; n = Nd * D * 4; }
    else if (nm == "node_out") { src = h->d_nout.p; n = Nd * 8 * 4; }
    else if (nm == "emb") { src = h->d_dbge.p; n = h->d_dbge.p ? Nd * D * 4 : 0; }
    else if (nm == "field") { src = h->d_dbgf.p; n = h->d_dbgf.p ? Nd * D * 4 : 0; }
    else if (nm == "tile_info") { src = h->d_tile.p; n = 64 * 4; }
    else if (nm == "dbg_w") { src = h->d_dbgw.p; n = h->d_dbgw.p ? E * WN * 4 : 0; }
    else if (nm == "phase_prof") { src = h->d_phase.p; n = h->d_phase.bytes; }
    else return fail(h, DEDF_ERR_INVALID, "unknown debug buffer " + nm);
    if (actual_bytes) *actual_bytes = n;
    if (!host_dst) return DEDF_OK;
    if (n > max_bytes) return fail(h, DEDF_ERR_INVALID, "destination too small");
    if (n && src) HIPCK(h, hipMemcpy(host_dst, src, n, hipMemcpyDeviceToHost));
    if (nm == "msg" && h->eo.msg_scale != 1.0f) {      // stored with a power-of-two scale (EdgeOffsets::msg_scale): the stage tests see the true message
        float* v = static_cast<float*>(host_dst);
        const float inv = 1.0f / h->eo.msg_scale;
        for (size_t i = 0; i < n / 4; ++i) v[i] *= inv;
    }
    return DEDF_OK;
}

int dedf_debug_packed(dedf_handle* h, const char* which, const float** ptr, size_t* n_floats) {
    if (!h || !which || !ptr || !n_floats) return DEDF_ERR_INVALID;
    const std::string w(which);
    const Image& im = w == "edge" ? h->edge_img : h->node_img;
    *ptr = im.data.data();
    *n_floats = im.data.size();
    return DEDF_OK;
}

// ---- per-node LayerNorm + LinearRS (no handle; current device; device pointers) --------------------------------------------------
int dedf_linear_rs_lmax(int lmax, const float* f, int n, const float* ln_w, const float* ln_b, const float* W, const float* bias, const int* valid,
                        float* out, void* stream) {
    if (!f || !W || !out || n <= 0 || ((ln_w == nullptr) != (ln_b == nullptr)) || (lmax != 2 && lmax != 3)) return DEDF_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int v0 = valid ? valid[0] : 0, v1 = valid ? valid[1] : 0, v2 = valid ? valid[2] : 0, v3 = valid && lmax == 3 ? valid[3] : 0;
    if (lmax == 3) {
        if (ln_w) hipLaunchKernelGGL((k_src_message<3, true>), dim3(n), dim3(64), 0, st, f, n, ln_w, ln_b, W, bias, out, v0, v1, v2, v3);
        else hipLaunchKernelGGL((k_src_message<3, false>), dim3(n), dim3(64), 0, st, f, n, ln_w, ln_b, W, bias, out, 0, 0, 0, 0);
    } else if (ln_w) hipLaunchKernelGGL((k_src_message<2, true>), dim3(n), dim3(64), 0, st, f, n, ln_w, ln_b, W, bias, out, v0, v1, v2, 0);
    else hipLaunchKernelGGL((k_src_message<2, false>), dim3(n), dim3(64), 0, st, f, n, ln_w, ln_b, W, bias, out, 0, 0, 0, 0);
    return hipGetLastError() == hipSuccess ? DEDF_OK : DEDF_ERR_RUNTIME;
}
int dedf_linear_rs(const float* f, int n, const float* ln_w, const float* ln_b, const float* W, const float* bias, const int* valid,
                   float* out, void* stream) {
    return dedf_linear_rs_lmax(2, f, n, ln_w, ln_b, W, bias, valid, out, stream);
}

// ---- graph primitives of the feature extractors (no handle; current device; device pointers) ------------------------------------
int dedf_fps(const float* x, int n, int n_samples, int start, int* idx_out, void* stream) {
    if (!x || !idx_out || n <= 0 || n_samples < 0 || n_samples > n || start < 0 || start >= n) return DEDF_ERR_INVALID;
    if (n > 64 * kFpsBlock) return DEDF_ERR_UNSUPPORTED;          // one workgroup keeps the cloud in registers: <= 65 536 points
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_samples == 0) return DEDF_OK;
    // 1 025 .. 16 384 points: Morton buckets + batches (dedf_graph.h::k_fps_bucketed<.., BATCH>): a sample only updates the buckets it can change,
    // and once the running minima form a plateau the samples are drawn 30-60 at a time by one wave from a candidate list, the buckets being
    // updated once per batch -- bit-identical to the exhaustive arg-max.  16 384 points, ratio 0.2: 6.3 ms exhaustive, 4.1 ms with buckets
    // (round 3), 1.5 ms with batches (round 6; profiles/r06p_*, r06r_*); it wins from ~1 100 points on.  DEDF_FPS_BUCKETED=0: the exhaustive kernel everywhere.
    // Below and above that range the exhaustive kernel: workgroup size by cloud size (tests/probe/fps_time.py: a sample costs a fixed part
    // that grows with the number of waves -- 0.43 / 0.79 / 1.95 us for 4 / 8 / 16 waves -- plus 0.04 us per point held by a thread).
    static const int bucketed = [] { const char* e = getenv("DEDF_FPS_BUCKETED"); return e ? atoi(e) : 1; }();
    if (bucketed != 0 && n <= 16384 && n > kFpsBatchedFrom && n_samples > kFpsBatchStart) {
        // (8 waves at every size: waves 0-3 draw the batch, one per SIMD, the other four share in collecting and applying it)
        if (n <= 4096) hipLaunchKernelGGL((k_fps_bucketed<8, 512, true>), dim3(1), dim3(512), 0, st, x, n, n_samples, start, idx_out);
        else if (n <= 8192) hipLaunchKernelGGL((k_fps_bucketed<16, 512, true>), dim3(1), dim3(512), 0, st, x, n, n_samples, start, idx_out);
        else hipLaunchKernelGGL((k_fps_bucketed<32, 512, true>), dim3(1), dim3(512), 0, st, x, n, n_samples, start, idx_out);
        return hipGetLastError() == hipSuccess ? DEDF_OK : DEDF_ERR_RUNTIME;
    }
    if (n <= 16 * 256) hipLaunchKernelGGL((k_fps<16, true, 256>), dim3(1), dim3(256), 0, st, x, n, n_samples, start, idx_out);
    else if (n <= 16 * 512) hipLaunchKernelGGL((k_fps<16, true, 512>), dim3(1), dim3(512), 0, st, x, n, n_samples, start, idx_out);
    else if (n <= 32 * 512) hipLaunchKernelGGL((k_fps<32, true, 512>), dim3(1), dim3(512), 0, st, x, n, n_samples, start, idx_out);
    else hipLaunchKernelGGL((k_fps<64, false>), dim3(1), dim3(kFpsBlock), 0, st, x, n, n_samples, start, idx_out);
    return hipGetLastError() == hipSuccess ? DEDF_OK : DEDF_ERR_RUNTIME;
}

size_t dedf_radius_scratch_bytes(int n_dst) { return n_dst > 0 ? ((size_t)n_dst + 1) * 8 + (size_t)n_dst * 4 : 0; }

int dedf_radius(const float* x_src, int n_src, const float* x_dst, int n_dst, float r, int max_num_neighbors, int exclude_self,
                int64_t edge_cap, int64_t* edge_dst, int64_t* edge_src, int64_t* n_edges, void* scratch, size_t scratch_bytes, void* stream) {
    if (!x_src || !x_dst || n_src <= 0 || n_dst <= 0 || !(r > 0.0f) || max_num_neighbors <= 0 || !n_edges || edge_cap < 0) return DEDF_ERR_INVALID;
    if (edge_cap > 0 && (!edge_dst || !edge_src)) return DEDF_ERR_INVALID;
    if (!scratch || scratch_bytes < dedf_radius_scratch_bytes(n_dst) || (reinterpret_cast<uintptr_t>(scratch) & 7)) return DEDF_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // caller-owned scratch (device): [n_dst + 1] int64 offsets (the last one = the total) | [n_dst] int32 counts.  No state in the library.
    int64_t* off = static_cast<int64_t*>(scratch);
    int* cnt = reinterpret_cast<int*>(off + n_dst + 1);
    const float r2 = r * r;
    const int nblk = (n_dst + kRadBlock / 64 - 1) / (kRadBlock / 64);          // one wave per destination
    int64_t* total = off + n_dst;
    hipLaunchKernelGGL(k_radius<false>, dim3(nblk), dim3(kRadBlock), 0, st, x_src, n_src, x_dst, n_dst, r2, max_num_neighbors, exclude_self,
                       cnt, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, cnt, n_dst, off, total);
    if (hipMemcpyAsync(n_edges, total, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return DEDF_ERR_RUNTIME;
    if (*n_edges > edge_cap) return DEDF_ERR_INVALID;            // *n_edges holds the size the caller must provide
    if (*n_edges > 0)
        hipLaunchKernelGGL(k_radius<true>, dim3(nblk), dim3(kRadBlock), 0, st, x_src, n_src, x_dst, n_dst, r2, max_num_neighbors, exclude_self,
                           cnt, off, edge_dst, edge_src);
    if (hipStreamSynchronize(st) != hipSuccess) return DEDF_ERR_RUNTIME;
    return DEDF_OK;
}

}  // extern "C"
