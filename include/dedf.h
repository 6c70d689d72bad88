/*
 * dedf.h — C ABI of the MI355X-native Diffusion-EDF score-head hot path (libdedf.so, gfx950).
 *
 * The reference (tomato1mule/diffusion_edf) is pure Python: it has no FFI/operator registry, and the boundary
 * of this path is the torch.nn.Module contract of ScoreModelHead / ScoreModelBase.  Each entry point below
 * replaces one piece of that contract (file:line under /root/reference) so that a maintainer can bind it with
 * ctypes (diffusion_edf_amd/_lib.py is that binding):
 *
 *   dedf_create          ScoreModelHead.__init__                 diffusion_edf/score_head.py:32-140
 *                        (+ MultiscaleTensorField.__init__       diffusion_edf/multiscale_tensor_field.py:22-190,
 *                           EquiformerBlock.__init__             diffusion_edf/gnn_block.py:71-162,
 *                           GraphAttentionMLP2.__init__          diffusion_edf/graph_attention.py:139-214)
 *   dedf_set_key_clouds  the `key_pcd_multiscale` argument        score_head.py:143; also hoists the pose-independent
 *                        prenorm_src + linear_src                 gnn_block.py:170-171
 *   dedf_set_key_weights `key_pcd_multiscale[n].w` (point attention)   gnn_block.py:190-194
 *   dedf_set_query       the `query_pcd` argument                 score_head.py:144,151-157
 *   dedf_score           ScoreModelHead.forward                   score_head.py:142-211
 *   dedf_sample          ScoreModelBase.sample (inner loop)       score_model_base.py:110-204
 *   dedf_set_radial_table   switch of dedf_sample's per-step radial table (no reference counterpart; see its declaration)
 *   dedf_linear_rs       LinearRS / ProjectIfMismatch per node     equiformer/tensor_product_rescale.py:176-185, skip.py:13-34
 *   dedf_fps, dedf_radius   torch_cluster fps / radius / radius_graph as connectivity.py:22,43,62 calls them
 *   dedf_energy          EbmScoreModelHead.compute_energy         score_head_ebm.py:122-174
 *   dedf_layer_forward   one {radial, gnn} layer of UnetFeatureExtractor      unet_feature_extractor.py:141-202, 289-324; block.py:141-174
 *   dedf_field           MultiscaleTensorField.forward at given points          multiscale_tensor_field.py:192-260 (KeypointExtractor's two fields,
 *                                                                                keypoint_extractor.py:179-188)
 *   dedf_keypoint_weight the weight head of KeypointExtractor                   keypoint_extractor.py:113-119, 189-194; gnn_block.py:112,214-216
 *   dedf_destroy         module deletion
 *
 * Conventions: all tensor arguments are DEVICE pointers owned by the caller (the library never frees or mutates
 * inputs); row-major; poses are (nT,7) = [qw,qx,qy,qz,x,y,z] in centimetres (reference README.md:77-82); features
 * are (N, sum_l mul_l (2l+1)) with irreps blocks 0e|1e|2e, mul-major, m fastest (reference layer_norm.py:111).
 * Every call enqueues work on `stream` (a hipStream_t; NULL = default stream) and returns without synchronising,
 * except where stated.  A handle is not thread-safe; use one handle per (device, stream).  Functions return 0 on
 * success or a DEDF_ERR_* code; dedf_last_error() gives the message.
 */
#ifndef DEDF_H
#define DEDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEDF_OK 0
#define DEDF_ERR_INVALID 1      /* bad argument / shape (the reference raises AssertionError / ValueError) */
#define DEDF_ERR_UNSUPPORTED 2  /* configuration outside the accelerated path */
#define DEDF_ERR_RUNTIME 3      /* HIP failure, workspace overflow, non-finite score */
#define DEDF_MAX_SCALES 8

typedef struct dedf_handle dedf_handle;

typedef struct dedf_config {
    int lmax;                            /* irreps = 64x0e + 32x1e (+ 16x2e (+ 8x3e)); SH 0..lmax.  Supported: 1, 2, 3.  (lmax 3: the kernels run the 8x3e block
                                            zero-padded as 16x3e -- an exact embedding, see diffusion_edf_amd/csrc/dedf_net.h; this interface keeps the
                                            reference's shapes: features are (N, 296), parameters as the reference's state dict has them) */
    int mul[4];                          /* must equal {64,32,16,8}[0..lmax] (every reference config) */
    int num_heads;                       /* 4 */
    int fc_neurons[3];                   /* {64 + time_emb_mlp[2],128,64} ({64,128,64} for the EBM head and for the score head WITHOUT edge time
                                            encoding -- ebm = 0, query_time_encoding = 1, the reference constructor's default), {128,32,32} (sapien place_* score heads), or
                                            {64,32,32} with ebm = 1 (the context-free fields of KeypointExtractor, lmax 2 only):
                                            fc_neurons after the -1 is resolved (multiscale_tensor_field.py:63-67) */
    int length_emb_dim;                  /* 64 */
    int time_emb_mlp[3];                 /* {256,128,64}, or {512,256,128} (sapien high-res configs; fc_neurons[0] = 192) */
    int irreps_mlp_mid;                  /* 3 */
    int n_scales;                        /* len(r_cluster_multiscale) */
    float radii[DEDF_MAX_SCALES];        /* r_cluster_multiscale; <= 0 means None (all-pairs scale) */
    float r_mincut_nonscalar_sh;         /* 0.3 */
    float length_enc_max_r;              /* 100 */
    float max_time;                      /* 1 */
    float time_enc_n;                    /* 10000 */
    float lin_mult, ang_mult;            /* 15, 2.5 */
    int max_neighbors;                   /* 1000 (multiscale_tensor_field.py:195) */
    int device;                          /* HIP device ordinal; -1 = host-only handle (packing tests, no GPU calls) */
    int64_t max_edges;                   /* edge workspace capacity per call; 0 = auto */
    int ebm;                             /* 1: EbmScoreModelHead (score_head_ebm.py): energy critic, no time encoding, fc_neurons[0] = 64,
                                            no lin/ang_vel_tp parameters; only dedf_energy is available.  0: ScoreModelHead */
    int half_gemm;                       /* 1: the reference's half_precision knob (agent.py:29,50-51 `model.half()`): every GEMM as ONE fp16 MFMA product
                                            (fp16 operands, fp32 accumulate) instead of the 3-term split; everything else stays fp32.  0: default */
    int use_src_point_attn;              /* 1: PointAttentiveScoreModel (point_attentive_score_model.py:71-72): the attention of every edge is
                                            multiplied, after the softmax, by its key point's weight (gnn_block.py:190-194,
                                            graph_attention.py:257-258); the weights come through dedf_set_key_weights.  0: default */
    int unet_layer;                      /* 1: the handle is ONE layer of the UNet feature extractor (unet_feature_extractor.py:141-202):
                                            GaussianRadialBasisLayerFiniteCutoff + block.EquiformerBlock with GraphAttentionMLP, irreps_src =
                                            irreps_dst = 64x0e+32x1e+16x2e, fc_neurons {64,32,32} (levels 2, 3 and the mid block of every shipped
                                            UNet config); radii[0] = the level's connection radius, n_scales = 1; the time / score fields are
                                            ignored; only dedf_layer_forward is available; half_gemm applies.  0: score / critic head */
    int unet_valid[4];                   /* UNet layer: true multiplicities of the block's irreps (0 = all of mul[]).  A 32x0e+16x1e+8x2e layer (levels 0-1 of
                                            the shipped UNets) runs on the 64/32/16 kernels with zero-padded parameters (diffusion_edf_amd/unet.py
                                            builds them); what the padding cannot express -- LayerNorm statistics over the true channels only -- is told
                                            to the kernels by these counts */
    int unet_fc_valid[3];                /* UNet layer: true {num_basis, h1, h2} of the radial network when narrower than fc_neurons (0 = fc_neurons) */
    int unet_narrow;                     /* UNet layer, 1: BOTH node sets carry the narrow level shape 32x0e+16x1e+8x2e(+4x3e) (levels 0-1 of the panda UNets,
                                            configs/panda_mug/pick_lowres/score_model_configs.yaml:33-55) with every true channel c of a block of m in M kernel
                                            channels at head * (M / 4) + 4 * (k / q) + k % q, k = c % (m / 4), q = 4 m / M (diffusion_edf_amd/unet_pad.py::place):
                                            the narrow instantiations of the layer kernels skip the lane-local work on the structurally zero channels.
                                            Requires unet_valid = {32, 16, 8 (, 4)}.  0: any zero-padded embedding whose padded channels are exact zeros (the
                                            padding is computed like data) -- EXCEPT the l = 3 block of an lmax-3 layer: every lmax-3 kernel skips the channels
                                            p with p % 4 >= 2 of the 16x3e block, so the (at most 8) true 3e channels MUST sit at the positions of
                                            unet_pad.py::place (8x3e: 0, 1, 4, 5, 8, 9, 12, 13; 4x3e: 0, 4, 8, 12) whatever this flag says.  dedf_create checks it (round 6) on the
                                            tensors whose l = 3 block is addressable by name -- linear_src / linear_dst / proj, norm_2 -- and returns
                                            DEDF_ERR_INVALID for a blob with a non-zero weight on a skipped channel */
    int query_time_encoding;             /* 1: ScoreModelHead(query_time_encoding=True) (score_head.py:64-70, 168-173): the query points carry
                                            query_time_mlp(time) -- time_emb_mlp[2] scalars -- as the DESTINATION feature of the key field's block
                                            (use_dst_feature, gnn_block.py:109-130, 170-180, 205-206): LayerNorm + LinearRS(bias) of it joins every edge's
                                            message, linear_src has no bias, and its projection skip_1 joins the attention output before post_norm.  The
                                            parameter list gains query_time_mlp.*, ...prenorm_dst.*, ...linear_dst.*, ...skip_1.skip.* and loses
                                            ...linear_src.bias.0.  Together with edge time encoding (fc_neurons[0] = 64 + time_emb_mlp[2]): instantiated
                                            for lmax 2 with fc_neurons {128,128,64} (half_gemm too) / {128,32,32} / {192,128,64} and lmax 1 / 3 with {128,128,64}.
                                            ALONE (edge_time_encoding=False, score_head.py:183-186: the pre-linears see the length embedding only,
                                            fc_neurons[0] = 64): instantiated for lmax 1-3 with fc_neurons {64,128,64}, full precision.
                                            0: default (every shipped config) */
} dedf_config;

typedef struct dedf_schedule {           /* host arrays, one entry per denoising step (score_model_base.py:146-171) */
    int n_steps;
    const double* t;                     /* diffusion time of the step */
    const double* alpha_ang;             /* ang_mult^2 * t^e_alpha * dt */
    const double* alpha_lin;             /* lin_mult^2 * t^e_alpha * dt */
    const double* temperature;           /* T_base * t^e_T */
} dedf_schedule;

typedef struct dedf_stats {
    int64_t n_dst;                       /* nT * nQ of the last call */
    int64_t n_edges[DEDF_MAX_SCALES];    /* edges per scale of the last score evaluation */
    int64_t n_edges_total;
    int overflow;                        /* 1 if the edge workspace was too small in any evaluation of the last dedf_score / dedf_energy /
                                            dedf_sample call (its outputs are NaN from that evaluation on; dedf_sample returns DEDF_ERR_RUNTIME) */
    int nonfinite;                       /* 1 if a score / energy of the last call was not finite: an activation left the fp16 operand range
                                            of the split-fp16 GEMMs, or inputs were not finite (dedf_sample returns DEDF_ERR_RUNTIME) */
    float rtab_err[DEDF_MAX_SCALES];     /* accuracy guard of the sampler's radial table (dedf_set_radial_table), last dedf_sample call that used it: per
                                            scale the largest |interpolated - exact| activation over ALL interval midpoints of the grid, checked at the
                                            call's first, middle and last diffusion time */
    int rtab_fallback;                   /* bit n set: scale n exceeded its bound (finite scales 1e-5, DEDF_RADIAL_TABLE_BOUND; all-pairs scale 2.4e-4, see
                                            dedf_set_radial_table) and evaluated its front per edge */
    int sample_retries;                  /* ABI 5: how often the last dedf_sample call was repeated with a doubled automatic edge workspace (0-3); a
                                            2-4x longer call is visible here */
    int edges_per_dst_capacity;          /* ABI 5: edges per destination node the automatic workspace is sized for after that (96 or more); grown by
                                            dedf_sample's repeats, kept for later calls on the same scene, reset by dedf_set_key_clouds, put back to its
                                            value before the call when every repeat failed.  0 with an explicit dedf_config.max_edges */
} dedf_stats;

const char* dedf_version(void);

/* ABI version of this header: bumped whenever a struct below changes its size or field order, or an entry point its arguments.  A binding
 * built against another version must not call into the library: dedf_get_stats / dedf_profile_read write sizeof(struct) bytes into caller
 * memory.  dedf_abi_version() is what the LOADED library was built with; dedf_struct_size(which) gives its sizeof of the struct
 * dedf_config [which = 0], dedf_schedule [1], dedf_stats [2], dedf_profile [3] for bindings that mirror the structs by hand (diffusion_edf_amd/_lib.py checks both).
 *   1  rounds 1-2.   2  round 3: dedf_config.unet_valid[4] (was [3]), dedf_stats.rtab_err / rtab_fallback, dedf_radius scratch arguments.
 *   3  round 4: dedf_abi_version / dedf_struct_size themselves; no struct change.   4  round 4: dedf_config.unet_narrow (appended).
 *   5  round 5: dedf_stats.sample_retries / edges_per_dst_capacity (appended).   6  round 5: dedf_config.query_time_encoding (appended). */
#define DEDF_ABI_VERSION 6
int dedf_abi_version(void);
size_t dedf_struct_size(int which);

/* canonical parameter order = order of the flat `params` blob of dedf_create; names are the reference state_dict keys
 * below `score_head.` (reference trainer.py:141-147) */
int dedf_param_count(const dedf_config* cfg);
const char* dedf_param_name(const dedf_config* cfg, int i);
size_t dedf_param_numel(const dedf_config* cfg, int i);

/* `params`: HOST pointer, concatenation of all tensors in canonical order, n_params floats in total. */
int dedf_create(const dedf_config* cfg, const float* params, size_t n_params, dedf_handle** out);
void dedf_destroy(dedf_handle* h);
const char* dedf_last_error(const dedf_handle* h);

/* x[s]: (n_pts[s],3), f[s]: (n_pts[s],D) device pointers; host arrays of pointers.  Copies the clouds and precomputes
 * the source message.  Synchronises `stream` before returning. */
int dedf_set_key_clouds(dedf_handle* h, int n_scales, const int* n_pts, const float* const* x, const float* const* f, void* stream);
/* w[s]: (n_pts[s],) device pointers: the key points' attention weights `FeaturedPoints.w` (gnn_block.py:191-193).  Required
 * before scoring when dedf_config.use_src_point_attn = 1; call after dedf_set_key_clouds (same n_pts). */
int dedf_set_key_weights(dedf_handle* h, int n_scales, const int* n_pts, const float* const* w, void* stream);
/* x: (nQ,3), f: (nQ,D), w: (nQ,) device pointers (copied). */
int dedf_set_query(dedf_handle* h, int nQ, const float* x, const float* f, const float* w, void* stream);

/* Ts: (nT,7) f32, time: (nT,) f32 -> ang: (nT,3), lin: (nT,3) f32 ("ang first", score_head.py:211). */
int dedf_score(dedf_handle* h, int nT, const float* Ts, const float* time, float* ang, float* lin, void* stream);

/* EbmScoreModelHead.compute_energy (score_head_ebm.py:122-174; used by agent.py:163-174 to rank the sampled poses):
 * Ts (nT,7) f32, time (nT,) f32 (ignored by the shipped critic configs: no time encoding) -> energy (nT,) f32.  EBM handles only. */
int dedf_energy(dedf_handle* h, int nT, const float* Ts, const float* time, float* energy, void* stream);

/* T_seed: (nT,7) f64; noise: NULL (counter-based Philox keyed by (seed, first_pose_index + pose, step)) or
 * (n_steps,2,nT,3) f64 standard normals [ang, lin]; Ts_out: (n_steps + 2, nT, 7) f64 = [seed, after each step, final again]
 * (score_model_base.py:199-201).  Synchronises `stream` before returning (overflow check). */
int dedf_sample(dedf_handle* h, int nT, const double* T_seed, const dedf_schedule* sched, uint64_t seed,
                int64_t first_pose_index, const double* noise, double* Ts_out, void* stream);

/* One UNet layer (dedf_config.unet_layer = 1) on a bipartite graph, reference unet_feature_extractor.py:289-302 (pool layer) / :316-324
 * (radius-graph layers) -> block.EquiformerBlock.forward (block.py:141-174):
 *   x_src (n_src,3), f_src (n_src,D), x_dst (n_dst,3), f_dst (n_dst,D) f32; edge_src / edge_dst (n_edges,) int64 as torch_cluster /
 *   dedf_radius return them, SORTED BY edge_dst (dedf_radius, FpsPool and RadiusGraph order); out (n_dst,D) f32 = the block's output
 *   features of the destination nodes.  Nodes without incoming edges get the bias-only attention output, like the reference's scatter.
 * Returns DEDF_ERR_INVALID if edge_dst is not sorted or an index is out of range (checked on device, reported after a sync). */
int dedf_layer_forward(dedf_handle* h, int n_src, const float* x_src, const float* f_src, int n_dst, const float* x_dst, const float* f_dst,
                       int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst, float* out, void* stream);

/* The sampler's radial table.  When every pose of a launch shares the diffusion time (dedf_sample; dedf_score never does: it takes one time per
 * pose), everything in front of layer 3 of the radial network -- length encoding, edge pre-linear with the time embedding, RadialProfile layers
 * 1 and 2 with their LayerNorm + SiLU (multiscale_tensor_field.py:225-234, equiformer/radial_func.py:11-60) -- depends on (scale, edge length)
 * only.  By default dedf_sample evaluates it once per step on a fine length grid per scale (2 048 intervals over [0, r) for a finite scale,
 * 16 384 over [0, 1.5 length_enc_max_r) for the all-pairs scale; longer edges are evaluated per edge) with the edge kernel's own code, and the
 * edge kernel interpolates the 64 activations per edge (4-point Lagrange; measured deviation from the per-edge evaluation: see DESIGN.md section 5).
 * Batches below 8 192 pose x query nodes evaluate per edge as well (the generator launch would cost more than it saves).
 * on = 1 is this default; on = 0 restores the per-edge evaluation everywhere, on = 2 uses the table at every batch size (tests) -- also through the
 * environment: DEDF_RADIAL_TABLE=0|1|2.  Instantiated for the lmax-2 score heads of the shipped
 * configs in full precision -- fc_neurons {128,128,64}, {192,128,64} (time_emb_mlp {512,256,128}) and {128,32,32} -- and for lmax 3 {128,128,64};
 * lmax 1, the half-precision mode, the EBM critic and the UNet layers evaluate per edge.
 * Accuracy guard: the length encoder's widths are TRAINABLE (radial_func.py:208-227: std = softplus(std_logit) + 1e-5), so a fixed grid is not
 * accurate for every checkpoint.  Every dedf_sample call that uses the table first evaluates the front exactly at the midpoint of EVERY grid
 * interval of every scale (at the call's first, middle and last diffusion time) and compares it with the interpolated table; a scale whose largest
 * deviation exceeds 1e-5 (absolute, on the O(1) activations; measured at the init widths: 2e-6, at sigma = 3e-3 r: 3e-3, i.e. 1e-4 of the score)
 * evaluates its front per edge for that call: dedf_stats.rtab_err / rtab_fallback.  The all-pairs scale (parameter-free sinusoidal encoder) is
 * held to two fp32 ulps of its largest sin / cos argument instead (2.4e-4): that rounding is in the per-edge evaluation just the same. */
int dedf_set_radial_table(dedf_handle* h, int on);

/* Chains of layers (a whole UNet is 17 of them): with on = 1, dedf_layer_forward returns WITHOUT synchronising; the verdict of its edge-list
 * check accumulates on the device and is returned -- and cleared -- by dedf_layer_check, which synchronises `stream`: DEDF_OK or
 * DEDF_ERR_INVALID.  Bad edges are replaced by (0, 0) inside the call, so a deferred verdict never means an out-of-range access. */
int dedf_layer_defer_check(dedf_handle* h, int on);
int dedf_layer_check(dedf_handle* h, void* stream);

/* The layers of one extractor run one after the other, so they can work in ONE per-call workspace (source / destination messages, int32 edge
 * lists, segment records E x 244 floats, aggregate, status words) instead of one each: after dedf_layer_share_workspace(h, owner) the calls of
 * `h` use the workspace and the deferred verdict word of `owner` (another UNet-layer handle on the same device; owner = NULL or h detaches),
 * and h's own per-call buffers are released.  A 4-scale UNet has 17-25 layers; on the 16 384-point scene their own workspaces add up to ~2 GB
 * that are used one at a time.  Contract: calls of handles that share a workspace are issued on ONE stream (or otherwise ordered), `owner`
 * outlives the link (detach before dedf_destroy(owner)), and one dedf_layer_check -- on any of them -- collects the verdict of all. */
int dedf_layer_share_workspace(dedf_handle* h, dedf_handle* owner);

/* MultiscaleTensorField.forward(query_points, input_points_multiscale, context_emb=None) (multiscale_tensor_field.py:192-260) for a field
 * without context encoding and without query features -- the key field of an EBM-type handle (dedf_config.ebm = 1), which is also what
 * KeypointExtractor.tensor_field / .weight_field are (keypoint_extractor.py:97-112: irreps_query = None, edge_context_emb_dim = None).
 * x: (n,3) points in the frame of the key clouds (dedf_set_key_clouds must have been called) -> field_out (n,D): the block's output
 * FFN(post_norm(emb)) + emb, and, if emb_out != NULL, emb (n,D): the attention output before the FFN (gnn_block.py:199-209) -- a field whose
 * irreps_output differs from its input irreps ends in skip_2(emb) instead of emb, see dedf_keypoint_weight.  Reference feature layout.
 * Replaces the query cloud of the handle (call dedf_set_query again before dedf_energy).  Points without any neighbour get the bias-only value. */
int dedf_field(dedf_handle* h, int n, const float* x, float* field_out, float* emb_out, void* stream);

/* The scalar head of KeypointExtractor.weight_field + weight_post (keypoint_extractor.py:111-119,185-194) from the outputs of dedf_field on a handle
 * whose ffn.fctp_2 is the weight field's (192x0e -> 64x0e; the rows of the other degrees zero):
 *   pre = (field[:, :64] - emb[:, :64]) + skip_W^T emb[:, :64] + skip_b         FFN output + skip_2 = LinearRS(emb -> 64x0e, bias), gnn_block.py:112,214-216
 *   w   = act(lin_w . SiLU(LayerNorm(pre; ln_w, ln_b)) + lin_b) * mult          torch.nn.LayerNorm(64) eps 1e-5, SiLU, Linear(64,1); act = sigmoid (1) or identity (0)
 * field, emb: (n,D) with D >= 64 = row stride `stride` floats; skip_W (64,64) [in][out]; all device pointers except lin_b, mult; out (n,). */
int dedf_keypoint_weight(const float* field, const float* emb, int n, int stride, const float* skip_W, const float* skip_b, const float* ln_w,
                         const float* ln_b, const float* lin_w, float lin_b, int sigmoid, float mult, float* out, void* stream);

int dedf_get_stats(dedf_handle* h, dedf_stats* out);   /* synchronises the last used stream */

/* Live kernel timing for bench.py's roofline: when enabled, HIP events are recorded on the launch stream around every
 * kernel class of every score evaluation.  dedf_profile_read synchronises, returns the totals since the last reset and resets. */
#define DEDF_PROF_CLASSES 6   /* 0 pose+time, 1 neighbour search, 2 fused edge kernel, 3 softmax-aggregate, 4 node kernel, 5 reduce */
typedef struct dedf_profile {
    int64_t n_evals;                       /* score evaluations covered */
    int64_t n_edges;                       /* sum of edge counts over those evaluations */
    int64_t n_dst;                         /* sum of destination-node counts */
    double ms[DEDF_PROF_CLASSES];          /* summed kernel time per class [ms] */
} dedf_profile;
int dedf_profile_enable(dedf_handle* h, int on);
int dedf_profile_read(dedf_handle* h, dedf_profile* out);

/* Test hooks: copy an internal device buffer of the last dedf_score call to HOST memory.  Names: "msg", "qpos", "pose",
 * "tb", "edge_src", "edge_dst", "edge_out", "z", "node_out", "tile_info", "dbg_w" (enable with dedf_debug_enable). */
int dedf_debug_enable(dedf_handle* h, int on);
int dedf_debug_copy(dedf_handle* h, const char* name, void* host_dst, size_t max_bytes, size_t* actual_bytes);
/* Packed weight images (host-only handles too): "edge" | "node"; for layout tests. */
int dedf_debug_packed(dedf_handle* h, const char* which, const float** ptr, size_t* n_floats);

/* ---- Graph primitives of the feature extractors (SURVEY 8(f) row 1 building blocks; no handle, current device) ----------------
 * They replace the torch_cluster calls of reference diffusion_edf/connectivity.py (torch_cluster is an un-vendored dependency;
 * the semantics are restated in oracle/graph_oracle.py).  Single cloud (all batch indices 0, as in every shipped config).
 *
 * dedf_fps     fps(src, ratio, random_start=False)  connectivity.py:62 : x (n,3) f32 -> idx_out (n_samples,) i32 in selection
 *              order, first = `start` (0 for random_start=False), n_samples = ceil(ratio * n); n <= 65 536.
 * dedf_radius  radius(x, y, r, max_num_neighbors)   connectivity.py:43 / radius_graph(x, r, loop=False, ...) :22 (exclude_self = 1,
 *              x_dst = x_src): all (dst, src) with |x_dst - x_src| < r, at most max_num_neighbors per dst (the first in source
 *              order), sorted by dst then src, as int64 like torch.  *n_edges (HOST) receives the edge count; if it exceeds
 *              edge_cap nothing is written and DEDF_ERR_INVALID is returned (call again with room).  Synchronises `stream`. */
/* dedf_linear_rs  per-node [EquivariantLayerNormV2 +] LinearRS on 64x0e+32x1e+16x2e features (skip.py:13-34 ProjectIfMismatch, the UNet's
 *              input / output projections): f (n,240) -> out (n,240).  W: per degree l a (mul_l x mul_l) [in][out] matrix, concatenated;
 *              bias (64) or NULL; ln_w (112) / ln_b (64) or both NULL (no LayerNorm); valid[3]: true multiplicities for the LayerNorm
 *              statistics (padded models), NULL = full.  All device pointers. */
int dedf_linear_rs(const float* f, int n, const float* ln_w, const float* ln_b, const float* W, const float* bias, const int* valid,
                   float* out, void* stream);
/* the same for the kernel layout of `lmax` (2: 240 floats per node as above; 3: 352 = 64x0e+32x1e+16x2e+16x3e, the UNet's lmax-3 layers;
 * ln_w (128), valid[4]) */
int dedf_linear_rs_lmax(int lmax, const float* f, int n, const float* ln_w, const float* ln_b, const float* W, const float* bias, const int* valid,
                        float* out, void* stream);
int dedf_fps(const float* x, int n, int n_samples, int start, int* idx_out, void* stream);
/* scratch: caller-owned DEVICE memory of at least dedf_radius_scratch_bytes(n_dst) bytes, 8-byte aligned (counts and offsets of the two
 * passes): the library keeps no state between calls */
size_t dedf_radius_scratch_bytes(int n_dst);
int dedf_radius(const float* x_src, int n_src, const float* x_dst, int n_dst, float r, int max_num_neighbors, int exclude_self,
                int64_t edge_cap, int64_t* edge_dst, int64_t* edge_src, int64_t* n_edges, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEDF_H */
