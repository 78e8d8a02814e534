/*
 * fcflow.h — C ABI of the MI355X-native FlowCompare forward log-prob engine (libfcflow.so).
 *
 * The reference (SamGalanakis/FlowCompare) has NO FFI seam on this path: the seam is Python
 * (`models_dict` + `inner_loop`, model_initialization.py:206-228).  This header is therefore the
 * boundary a maintainer would bind with ctypes; each entry point names the
 * reference interface it replaces.  Conventions:
 *   - plain C, opaque handles, caller-owned DEVICE buffers, explicit HIP stream (void* = hipStream_t),
 *     no hidden allocation inside the compute calls (the caller provides the workspace);
 *   - every function returns an int status (FC_OK == 0); fc_last_error() gives the message.
 *     Nothing in the library calls exit()/abort() (the reference's vendored CUDA helpers do,
 *     lib/paconv_lib/src/gpu/cuda_utils.h:32-41 — deliberately not reproduced);
 *   - a handle is immutable after create; concurrent calls on one handle must use distinct
 *     workspaces; all tensors are fp32, contiguous, channels-last [B, points, features].
 *   - weights are handed over as the reference checkpoint's state_dict entries (name, shape, host
 *     pointer); folding / padding / packing for the kernels happens inside create.
 * ABI history: 1 = forward log-prob engine; 2 = + inverse / sampling, staging and change-map entries, profiler filter;
 * 3 = + the stateless training primitives fc_train_* (forward AND backward of every node of the path, no handles: parameters stay
 *     the caller's dense fp32 device tensors because they change every optimiser step);
 * 4 = + the PAConv embedder's training primitives (softmax / assign_score / centre difference / gathered-row gradients / 3-NN
 *     interpolation), LeakyReLU slope argument of fc_train_edge_fwd_f32 / fc_train_edge_bwd_prep_f32 (0 = ReLU), optimiser step
 *     (fc_train_sqnorm_f32, fc_train_adam_f32);
 * 5 = + the deferred range check (fc_range_check_defer / _resolve / _pending);
 * 6 = + fc_profile_stride (sampled bracketing of the in-library kernel timing), fc_train_linear_act_fwd_f32 / fc_train_linear_dgrad_act_f32 (activation and its backward in the GEMM epilogues).
 */
#ifndef FCFLOW_H
#define FCFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_ABI_VERSION 7

enum fc_status {
    FC_OK = 0,
    FC_ERR_INVALID = 1,      /* bad argument / inconsistent config            */
    FC_ERR_MISSING = 2,      /* a state_dict entry the config needs is absent */
    FC_ERR_SHAPE = 3,        /* a tensor has the wrong shape                  */
    FC_ERR_WORKSPACE = 4,    /* workspace too small                           */
    FC_ERR_HIP = 5,          /* HIP runtime error                             */
    FC_ERR_UNSUPPORTED = 6   /* valid reference config this build does not cover yet */
};

/* One state_dict entry in HOST memory.  Integer buffers (Permuter.permutation) are passed as floats. */
typedef struct fc_tensor {
    const char* name;
    const float* data;
    int32_t ndim;
    int64_t shape[4];
} fc_tensor;

/* FC_FLOW_EXPONENTIAL (models/exponential_coupling.py:44-58) -- PERMANENT CAP: latent_dim - latent_dim / 2 <= 16, inference and training alike
 * (fc_flow_create returns FC_ERR_UNSUPPORTED beyond it).  The layer's coupling net emits d2 * d2 + d2 numbers per point (a dense matrix per
 * point): at the reference's latent width of 300 that is 22 650 outputs and 90 KB of matrix per point and layer, which the reference itself
 * never runs (no shipped configuration selects this flow_type; its own smoke configurations use single-digit latents).  The per-point
 * matrix exponential therefore lives in one lane's registers (csrc/misc.hip expm_coupling_kernel) and is not tiled. */
enum fc_flow_type { FC_FLOW_AFFINE = 0, FC_FLOW_SPLINE = 1, FC_FLOW_EXPONENTIAL = 2 };
enum fc_scale_fn { FC_SCALE_EXP = 0, FC_SCALE_SIGMOID = 1 };
enum fc_act { FC_ACT_NONE = 0, FC_ACT_GELU = 1, FC_ACT_RELU = 2, FC_ACT_ELU = 3, FC_ACT_LRELU02 = 4 };
enum fc_permuter { FC_PERM_LINEAR_LU = 0, FC_PERM_RANDOM = 1, FC_PERM_FULL = 2, FC_PERM_EXPONENTIAL = 3 };
enum fc_expm_algo { FC_EXPM_TORCH = 0, FC_EXPM_ORIGINAL = 1 };

/* Scalar part of the reference config dict (config/<name>.yaml keys; layer widths are read from the tensor shapes). */
typedef struct fc_flow_config {
    int32_t struct_size;          /* sizeof(fc_flow_config), ABI check                         */
    int32_t input_dim;            /* 'input_dim' (6)                                           */
    int32_t latent_dim;           /* 'latent_dim'                                              */
    int32_t cif_latent_dim;       /* 'cif_latent_dim' (== latent_dim: plain PreConditionApplier) */
    int32_t n_flow_layers;        /* 'n_flow_layers'                                           */
    int32_t flow_type;            /* enum fc_flow_type  <- 'flow_type'                          */
    int32_t affine_scale_fn;      /* enum fc_scale_fn   <- 'affine_scale_fn'                    */
    int32_t permuter_type;        /* enum fc_permuter   <- 'permuter_type'                      */
    int32_t act_norm;             /* 'act_norm'                                                */
    int32_t nonlinearity;         /* enum fc_act        <- 'coupling_block_nonlinearity'        */
    int32_t global_context;       /* config['global'] (DGCNNembedderGlobal), model_initialization.py:42-45 */
    int32_t extra_context_dim;    /* config['extra_context_dim'] (0 or 1), model_initialization.py:33-39   */
    int32_t input_embedding_dim;  /* 'input_embedding_dim' (E)                                 */
    int32_t num_bins_spline;      /* 'num_bins_spline'                                         */
    int32_t expm_algo;            /* enum fc_expm_algo  <- 'coupling_expm_algo'                 */
    float linear_lu_eps;          /* 'linear_lu_eps'                                           */
    float eps_expm;               /* 'eps_expm'                                                */
    float clamp_dist;             /* 'clamp_dist' (CIF augment/slice std clamp)                */
} fc_flow_config;

typedef struct fc_flow fc_flow;      /* replaces models.Flow (models/transform.py:61-84)         */
typedef struct fc_dgcnn fc_dgcnn;    /* replaces models.DGCNNembedder(Global) (models/pytorch_gcn.py:50-188) */
typedef struct fc_paconv fc_paconv;  /* replaces models.PointNet2SSGSeg (models/scene_seg_PAConv/model/pointnet2/pointnet2_paconv_seg.py:14-82) */

int fc_abi_version(void);
const char* fc_last_error(void);     /* thread-local, valid until the next failing call on this thread */
/* Threading: handles are immutable after create, but a compute call uses the handle's range-guard word and the caller's
 * workspace, so concurrent calls must use different handles (or be serialised by the caller); different handles and the
 * fc_op_* / fc_stage_* / fc_change_* entry points may run concurrently on different streams.
 * Arithmetic: GEMM-shaped work carries fp32-equivalent operands as two fp16 limbs; a compute call whose
 * activations leave fp16's range (|x| >= 65504) is transparently repeated with bf16 limbs, so results stay fp32-accurate.  The
 * flow / embedder compute calls therefore end with a stream synchronise. */

/* ---- flow: Flow.log_prob / Flow.sample ------------------------------------------------------- */

/* Builds the device-resident, kernel-packed flow from the checkpoint tensors `flow.state_dict()`
 * (replaces the module graph initialize_flow assembles, model_initialization.py:136-160). */
int fc_flow_create(const fc_flow_config* cfg, const fc_tensor* tensors, int32_t n_tensors, fc_flow** out);
void fc_flow_destroy(fc_flow* flow);

/* Bytes of device workspace fc_flow_logprob_f32 / fc_flow_inverse_f32 need for (B, N, M). */
int fc_flow_workspace_bytes(const fc_flow* flow, int32_t B, int32_t N, int32_t M, size_t* bytes);

/* Number of noise tensors one forward consumes and the feature width of tensor i ([B,N,width]):
 * the augmenter's rsample() (models/augmenter.py:49-63) and one per CIF block (models/cif_block.py:74).
 * Making eps an explicit input is what makes the stochastic forward reproducible. */
int fc_flow_noise_count(const fc_flow* flow);
int fc_flow_noise_width(const fc_flow* flow, int32_t i);

/* Flow.log_prob (models/transform.py:70-76):
 *   x        [B,N,input_dim]        target points
 *   ctx      [B,M,E]                per-point context embedding (attention keys/values); in global
 *                                   mode the reference passes the embedding repeated to [B,N,E], so M == N
 *   extra    [B,X] or NULL          extra context, constant per scene (inner_loop repeats it over N)
 *   eps      n_eps device pointers  noise, eps[i] is [B,N,fc_flow_noise_width(i)]
 *   logprob  [B,N]       (out)      log p(x | ctx) in nats
 *   z_out    [B,N,latent_dim] or NULL (out) latent after the last transform (diagnostics / tests) */
int fc_flow_logprob_f32(fc_flow* flow, const float* x, const float* ctx, const float* extra,
                        const float* const* eps, int32_t n_eps, float* logprob, float* z_out,
                        int32_t B, int32_t N, int32_t M, void* workspace, size_t workspace_bytes, void* stream);

/* Inverse pass of Flow.sample (models/transform.py:79-84) from a caller-drawn latent z [B,N,latent_dim]
 * -> x_out [B,N,input_dim].  eps: one tensor per CIF block (Slice.inverse draws, models/slice.py:46-58). */
int fc_flow_inverse_f32(fc_flow* flow, const float* z, const float* ctx, const float* extra,
                        const float* const* eps, int32_t n_eps, float* x_out,
                        int32_t B, int32_t N, int32_t M, void* workspace, size_t workspace_bytes, void* stream);

/* ---- DGCNN context embedder: input_embedder(extract_0) --------------------------------------- */
int fc_dgcnn_create(int32_t n_neighbors, int32_t global_pool, const fc_tensor* tensors, int32_t n_tensors, fc_dgcnn** out);
void fc_dgcnn_destroy(fc_dgcnn* emb);
int fc_dgcnn_out_dim(const fc_dgcnn* emb);
int fc_dgcnn_workspace_bytes(const fc_dgcnn* emb, int32_t B, int32_t M, size_t* bytes);
/* pts [B,M,C_in] -> out [B,M,E] (per-point) or [B,E] (global_pool); eval-mode BatchNorm (running stats). */
int fc_dgcnn_embed_f32(fc_dgcnn* emb, const float* pts, float* out, int32_t B, int32_t M,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- PAConv context embedder: input_embedder(extract_0) for config['input_embedder'] == 'PAConv' ------------------ */
/* Weights = PointNet2SSGSeg.state_dict().  Includes native replacements of the six pointops_cuda kernels the reference uses on
 * this path (furthestsampling, gathering, knnquery_heap, grouping, nearestneighbor, interpolation:
 * models/scene_seg_PAConv/lib/pointops/src/pointops_api.cpp:16-40). */
int fc_paconv_create(const fc_tensor* tensors, int32_t n_tensors, fc_paconv** out);
void fc_paconv_destroy(fc_paconv* emb);
int fc_paconv_out_dim(const fc_paconv* emb);
int fc_paconv_workspace_bytes(const fc_paconv* emb, int32_t B, int32_t M, size_t* bytes);
/* pts [B,M,3+c] (xyz first) -> out [B,M,E]; needs M >= 256 (four 4x down-samplings); eval-mode BatchNorm. */
int fc_paconv_embed_f32(fc_paconv* emb, const float* pts, float* out, int32_t B, int32_t M,
                        void* workspace, size_t workspace_bytes, void* stream);
/* furthest point sampling alone (lib/pointops/src/sampling/sampling_cuda_kernel.cu:58-168): xyz [B,n,3] -> idx [B,m] int32.
 * Also the FPS subsampling step of the data loader that feeds the path. */
int fc_op_fps_f32(const float* xyz, int32_t* idx, int32_t B, int32_t n, int32_t m, void* stream);

/* ---- the steps either side of the path ------------------------ */
/* Farthest point subsampling of the voxel loader (dataloaders/ams_voxel_loader.py:298-307: torch_cluster.fps(x, batch=0,
 * ratio, random_start=False) on the FULL feature rows).  pts [B, n, ld] (first C <= 8 columns are the coordinates) ->
 * idx [B, m] int64 in selection order, idx[:,0] = 0, ties to the lowest index. */
int fc_stage_fps_f32(const float* pts, int32_t ld, int32_t C, int64_t* idx, int32_t B, int32_t n, int32_t m, void* stream);
/* Joint unit-sphere normalisation (utils.py:259-280 co_unit_sphere / unit_sphere): over cat(p0[b], p1[b]) subtract the xyz
 * mean, divide by the largest xyz norm; other columns pass through.  p0 [B,n0,ld], p1 [B,n1,ld] -> out0, out1;
 * inverse [B,4] = furthest_distance, mean x, y, z (the dict the reference returns with return_inverse=True). */
int fc_stage_co_unit_sphere_f32(const float* p0, int32_t n0, const float* p1, int32_t n1, int32_t ld, float* out0, float* out1,
                                float* inverse, int32_t B, void* stream);
/* Change map (test_flow.py:241-275 log_prob_to_change + clamp_infs): lp10 [B,N], lp00 [B,N0] are clamped IN PLACE when they
 * hold infs (all infs of a tensor -> its smallest non-inf entry), out [B,N] = 1 - (lp10 - min)/(max - min) where
 * lp10 < mean(lp00) - multiple * std(lp00) (per scene, unbiased std) or, with use_cutoff, lp10 < hard_cutoff; 0 elsewhere.
 * *invalid (host) is set to 1 when the result holds NaN / inf (the reference asserts is_valid). Synchronises the stream. */
/* clamp_infs alone (test_flow.py:241-247): every +-inf of t[0..n) becomes the smallest non-inf entry, in place. */
int fc_clamp_infs_f32(float* t, int64_t n, void* stream);
int fc_change_map_f32(float* lp10, int32_t N, float* lp00, int32_t N0, float* out, int32_t B, float multiple, float hard_cutoff,
                      int32_t use_cutoff, int32_t* invalid, void* stream);

/* ---- deferred range check ------------------------------------------------------------------------
 * The split-fp16 matrix loops cannot hold |activation| >= 65504: fc_flow_logprob_f32, fc_dgcnn_embed_f32 and
 * fc_paconv_embed_f32 run their fast pass, read one device flag back and, if it is set, repeat the whole pass on the unbounded-range
 * bf16-limb loops.  By default the read-back is a stream synchronisation inside the call.  With the calling thread's switch ON the call
 * only ENQUEUES the fast pass, a 4-byte copy of the flag into pinned host memory and an event, and returns: any number of forwards can be
 * queued back to back on one stream.  fc_range_check_resolve then visits the queued passes in order, waits for each pass's event and
 * repeats it if its flag is set (a pass behind a repeated one is repeated too: it may have consumed the other's output); *n_repeated (may be
 * NULL) receives the number of repeated passes.  Contract while passes are pending: their inputs, outputs and workspaces stay allocated
 * and are not written by the caller, and anything computed from their outputs is provisional until resolve has returned 0 repeats.
 * Switching the mode off resolves what is pending.  The state is per host thread. */
int fc_range_check_defer(int32_t on);
int fc_range_check_resolve(int32_t* n_repeated);
int32_t fc_range_check_pending(void);

/* ---- in-library kernel timing (used by bench.py for the roofline object) --------------------- */
/* When enabled, every kernel launch of this library is bracketed by HIP events on its launch stream.
 * fc_profile_report writes a JSON array [{"kernel", "launches", "ms", "flops", "bytes"}, ...] (kernel names as
 * rocprofv3 prints them; flops = useful multiply-add FLOPs excluding padding, bytes = algorithmic HBM bytes). */
int fc_profile_enable(int32_t on);
int fc_profile_reset(void);
/* Bracket only the launches whose kernel name contains kernel_substr (NULL or "" = all): two event records per launch cost
 * about 4 us of stream time each, 3 % of a C2 forward when every launch is bracketed. */
int fc_profile_filter(const char* kernel_substr);
/* Of the launches that pass the filter, bracket every n-th one only (n <= 1: all).  A launch-bound run (C1: ~940 launches of 11-16 us
 * per 13 ms forward, 575 of them the dominant kernel) is slowed by 30 % when each of those is bracketed; a sample of them gives the
 * same average duration.  Launches, ms, flops and bytes of fc_profile_report then count the bracketed launches only. */
int fc_profile_stride(int32_t n);
int fc_profile_report(char* buf, size_t cap);

/* ---- single operators (same kernels as above; exported for unit-level parity tests) ---------- */

/* y[rows,N] = act( x[rows,K] @ W[N,K]^T + bias + residual ), torch.nn.functional.linear semantics
 * (models/nets.py:19-30 building block).  bias / residual may be NULL.  All DEVICE pointers, dense row-major. */
int fc_op_linear_f32(const float* x, const float* W, const float* bias, const float* residual, float* y,
                     int32_t rows, int32_t N, int32_t K, int32_t act, void* stream);

/* The in_layer and hidden layers of one reference MLP (models/nets.py:19-30: act(in_layer(x)); even hidden layer i: keep = x, x = act(W x + b);
 * odd: x = act(keep + W x + b)) at hidden width 512 -- the coupling nets of models/affine_coupling.py:30-46 / models/spline_coupling.py:187-210
 * -- over cat(x0 [rows,k0], x1 [rows,k1]) (+ rowscal[row] * net.colvec, the rank-1 form of an extra-context column).  out [rows,512] = the last
 * hidden activation (what out_layer consumes).  use_rows != 0: the row-resident chain kernel (one launch, csrc/mlprows.hip), 0: one GEMM
 * launch per layer.  tensors are HOST fp32: net.in_layer.{weight,bias}, net.layers.<i>.{weight,bias}, net.out_layer.weight (shape check
 * only), optional net.colvec [512]; x0 / x1 / rowscal / out are DEVICE pointers (x1, rowscal may be NULL). */
int fc_op_mlp_hidden_f32(const float* x0, int32_t k0, const float* x1, int32_t k1, const float* rowscal, const fc_tensor* tensors,
                         int32_t n_tensors, float* out, int32_t rows, int32_t act, int32_t use_rows, void* stream);

/* out[B,N,D] = softmax(q k^T * scale) v  with q [B,N,D], k,v [B,M,D]  (models/perceiver.py:106-113). */
int fc_op_attention_f32(const float* q, const float* k, const float* v, float* out,
                        int32_t B, int32_t N, int32_t M, int32_t D, float scale, void* stream);

/* k nearest neighbours in feature space, reference ranking -|xi|^2 + 2 xi.xj - |xj|^2 (self included)
 * (models/pytorch_gcn.py:13-20).  f [B,M,C] channels-last -> idx [B,M,k] int32 (unordered set). */
int fc_op_knn_f32(const float* f, int32_t* idx, int32_t B, int32_t M, int32_t C, int32_t k, void* stream);
/* The same search started from given neighbour sets idx_warm [B,M,k] (may equal idx): what the DGCNN embedder does between its four
 * levels, which search one cloud in successive feature spaces (models/pytorch_gcn.py:43-60).  Any k distinct candidates bound the k-th
 * best from below, so the stream skips everything under that bound; the returned set is the exact top-k whatever idx_warm holds.  ABI v7. */
int fc_op_knn_warm_f32(const float* f, const int32_t* idx_warm, int32_t* idx, int32_t B, int32_t M, int32_t C, int32_t k, void* stream);

/* Elementwise rational-quadratic spline with linear tails (models/spline_coupling.py:24-169).
 * x [n], params [n, 3K+1] laid out [K widths | K heights | K+1 derivatives] -> y [n], logabsdet [n]. */
int fc_op_rqspline_f32(const float* x, const float* params, float* y, float* logabsdet,
                       int64_t n, int32_t K, int32_t inverse, void* stream);

/* ---- training primitives: backward of the hot path (first slice = the residual MLP, models/nets.py:19-30) ----
 * Replaces what torch.autograd does for torch.nn.Linear + activation inside MLP.forward when train.py:112 calls loss.backward().
 * Parameters stay dense fp32 [N, K] tensors owned by the caller (they change every optimiser step).  Activations are PANELS:
 * row-major fp32 [rows_pad, width_pad], rows_pad a multiple of 256, widths padded to 32 with ZERO pad columns, 16-byte aligned.
 * A Linear's input may be 1..3 panels side by side (seg_widths = their true widths, sum = K), e.g. cat(x1, context).
 * ovf: one device int32 the caller zeroes per optimisation step; non-NULL selects the split-fp16 MFMA loop and the primitives OR 1
 * into it when an operand leaves the fp16 range (results then invalid: repeat the step with ovf == NULL, the fp32-input MFMA loop). */
size_t fc_train_linear_pack_bytes(int32_t N, const int32_t* seg_widths, int32_t nseg);
/* W [N,K], bias [N] | NULL  ->  pack (256-byte aligned, fc_train_linear_pack_bytes long): operand images of W and W^T. */
int fc_train_linear_pack_f32(const float* W, const float* bias, int32_t N, const int32_t* seg_widths, int32_t nseg, void* pack,
                             size_t pack_bytes, int32_t* ovf, void* stream);
/* u [rows_pad, ldu] = cat(x...) W^T + bias (+ residual): the pre-activation, which the backward needs. */
int fc_train_linear_fwd_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* const* x, const int32_t* ldx,
                            int32_t rows_pad, const float* residual, int32_t ldr, float* u, int32_t ldu, int32_t* ovf, void* stream);
/* The same with the activation applied in the GEMM's epilogue: u (pre-activation, what the backward of the activation reads) and
 * y = act(u) (FC_ACT_GELU / _RELU / _ELU) from ONE launch, both with pitch ldu.  ABI v6. */
int fc_train_linear_act_fwd_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* const* x, const int32_t* ldx,
                                int32_t rows_pad, const float* residual, int32_t ldr, float* u, float* y, int32_t ldu, int32_t act, int32_t* ovf,
                                void* stream);
/* dx [rows_pad, lddx >= sum of padded segment widths] = du W  (columns in padded-segment order). */
int fc_train_linear_dgrad_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu,
                              int32_t rows_pad, float* dx, int32_t lddx, int32_t* ovf, void* stream);
/* Data gradient of a hidden Linear fused with the backward of the activation in front of it:
 * dx = (du . W + addend) * act'(u_prev), i.e. the gradient w.r.t. the previous layer's PRE-activation (addend: optional gradient of a
 * residual branch that joins at the previous layer's output; u_prev / addend / dx are [rows_pad, lddx] panels; one input segment).  ABI v6. */
int fc_train_linear_dgrad_act_f32(const void* pack, int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu, int32_t rows_pad,
                                  float* dx, int32_t lddx, const float* addend, const float* u_prev, int32_t act, int32_t* ovf, void* stream);
size_t fc_train_linear_wgrad_ws_bytes(int32_t N, const int32_t* seg_widths, int32_t nseg, int32_t rows);
/* dW [N,K] (=|+=) du[:rows]^T cat(x...)[:rows],  db [N] (=|+=) column sums of du[:rows]; either may be NULL.  fp32-input MFMA,
 * fixed summation order (bit-reproducible).  ws: 256-byte aligned scratch of fc_train_linear_wgrad_ws_bytes. */
int fc_train_linear_wgrad_f32(int32_t N, const int32_t* seg_widths, int32_t nseg, const float* du, int32_t ldu, const float* const* x,
                              const int32_t* ldx, int32_t rows, float* dW, float* db, int32_t accumulate, void* ws, size_t ws_bytes,
                              int32_t* ovf, void* stream);
/* y = act(u) and du = dy * act'(u) on dense [rows_pad, ld] panels (enum fc_act; GELU is the exact erf form);
 * rows >= `rows` of du are written as zeros. */
int fc_train_act_fwd_f32(const float* u, float* y, int32_t rows_pad, int32_t ld, int32_t act, void* stream);
int fc_train_act_bwd_f32(const float* dy, const float* u, float* du, int32_t rows_pad, int32_t rows, int32_t ld, int32_t act, void* stream);

/* Cross-attention core out = softmax(q k^T scale) v per scene (models/perceiver.py:106-113) on row-major matrices with pitches
 * (q, out, dout, dq: [B*N, ld]; k, v, dk, dv: [B*M, ld]; D = padded head dim, 32 or 64) and its backward, which recomputes the
 * scores tile by tile (nothing of [N, M] is stored).  fwd: ws (fc_train_attention_ws_bytes) + ovf select the split-fp16 kernel.
 * stats = device buffer of 2*B*N floats (log-sum-exp and dO.O per query): the split-fp16 forward leaves the log-sum-exp in its first
 * half and sets *stats_valid (host int), which the backward takes back to skip its own pass; ovf selects the split-fp16 kernels (D = 64). */
size_t fc_train_attention_ws_bytes(int32_t B, int32_t N, int32_t M, int32_t D);
int fc_train_attention_fwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, float* out, int32_t ldo,
                               int32_t B, int32_t N, int32_t M, int32_t D, float scale, void* ws, size_t ws_bytes, float* stats, int32_t* stats_valid,
                               int32_t* ovf, void* stream);
int fc_train_attention_bwd_f32(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* out,
                               int32_t ldo, const float* dout, int32_t lddo, float* dq, int32_t lddq, float* dk, int32_t lddk, float* dv,
                               int32_t lddv, float* stats, int32_t stats_valid, int32_t B, int32_t N, int32_t M, int32_t D, float scale, int32_t* ovf,
                               void* stream);

/* Rational-quadratic spline coupling element in the reference's parameter layout (models/spline_coupling.py:187-210: the coupling
 * MLP's output row is [d2][K width | K height | K+1 derivative logits]), forward (y2, ldj[row] = sum over dims of log|dy/dx|) and
 * the analytic backward w.r.t. x2 and every logit.  Pad columns of y2 / dx2 / dparams are written as zeros. */
int fc_train_rqspline_fwd_f32(const float* x2, int32_t ldx, const float* params, int32_t ldp, float* y2, int32_t ldy, float* ldj, int32_t rows,
                              int32_t d2, int32_t K, void* stream);
int fc_train_rqspline_bwd_f32(const float* x2, int32_t ldx, const float* params, int32_t ldp, const float* dy2, int32_t lddy, const float* dldj,
                              float* dx2, int32_t lddx, float* dparams, int32_t lddp, int32_t rows, int32_t d2, int32_t K, void* stream);
/* torch.nn.LayerNorm(width) of PreNorm (models/perceiver.py:18-27).  stats [2*rows] = (mean, rstd) per row, kept for the backward;
 * bwd writes dx and the panel dy*xhat, whose column sums are d gamma (d beta = column sums of dy): fc_train_colsum_f32. */
int fc_train_layernorm_fwd_f32(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy, float* stats,
                               int32_t rows, int32_t width, float eps, void* stream);
int fc_train_layernorm_bwd_f32(const float* x, int32_t ldx, const float* gamma, const float* dy, int32_t lddy, const float* stats, float* dx,
                               int32_t lddx, float* dy_xhat, int32_t ldt, int32_t rows_pad, int32_t rows, int32_t width, void* stream);
/* The per-element closures of the flow, forward and backward, one workgroup per point (pad columns written as zeros):
 *   affine  models/affine_coupling.py:23-46   st = [raw scale d2 | shift d2]: y2 = x2 s + t, ldj[row] = sum log s (scale_fn: enum fc_scale_fn)
 *   gauss   models/augmenter.py:49-63 + distributions.py:128-153   p = [mean nz | log std nz], eps [rows, nz] dense:
 *           z = mean + eps std, ldj[row] = -sum log N(z; mean, std); std = min(exp(log std), clamp) when clamp > 0 (CIF blocks)
 *   normlp  models/slice.py:31-44   out[row] = sum_j log N(v_j; mean_j, std_j), same parameter panel and clamp
 *   base    models/distributions.py:192-195   out[row] = sum over `width` columns of -x^2/2 - log(2 pi)/2 */
int fc_train_affine_fwd_f32(const float* x2, int32_t ldx, const float* st, int32_t ldst, float* y2, int32_t ldy, float* ldj, int32_t rows, int32_t d2,
                            int32_t scale_fn, void* stream);
int fc_train_affine_bwd_f32(const float* x2, int32_t ldx, const float* st, int32_t ldst, const float* dy2, int32_t lddy, const float* dldj, float* dx2,
                            int32_t lddx, float* dst, int32_t lddst, int32_t rows, int32_t d2, int32_t scale_fn, void* stream);
int fc_train_gauss_fwd_f32(const float* p, int32_t ldp, const float* eps, float* z, int32_t ldz, float* ldj, int32_t rows, int32_t nz, float clamp,
                           void* stream);
int fc_train_gauss_bwd_f32(const float* p, int32_t ldp, const float* eps, const float* dz, int32_t lddz, const float* dldj, float* dp, int32_t lddp,
                           int32_t rows, int32_t nz, float clamp, void* stream);
int fc_train_normlp_fwd_f32(const float* v, int32_t ldv, const float* p, int32_t ldp, float* out, int32_t rows, int32_t nz, float clamp, void* stream);
int fc_train_normlp_bwd_f32(const float* v, int32_t ldv, const float* p, int32_t ldp, const float* g, float* dv, int32_t lddv, float* dp, int32_t lddp,
                            int32_t rows, int32_t nz, float clamp, void* stream);
/* ExponentialCoupling element (models/exponential_coupling.py:44-58): o = [d2*d2 raw matrix | d2 shift] per point, scal4 = device
 * (scale, shift, rescale, reshift); y2 = expm(rescale tanh(scale raw + shift) + reshift + 1e-8) x2 + b, ldj = trace.  d2 <= 16.
 * status (device int32): set when a matrix norm exceeds the 64 squarings the backward keeps states for.
 * bwd: dscal [rows, 4] = per-point parts of the four scalars' gradients (column sums = the gradients). */
int fc_train_expm_fwd_f32(const float* x2, int32_t ldx, const float* o, int32_t ldo, const float* scal4, float* y2, int32_t ldy, float* ldj, int32_t rows,
                          int32_t d2, int32_t* status, void* stream);
int fc_train_expm_bwd_f32(const float* x2, int32_t ldx, const float* o, int32_t ldo, const float* scal4, const float* dy2, int32_t lddy, const float* dldj,
                          float* dx2, int32_t lddx, float* dout, int32_t lddo, float* dscal, int32_t rows, int32_t d2, void* stream);
int fc_train_base_fwd_f32(const float* x, int32_t ldx, float* out, int32_t rows, int32_t width, void* stream);
int fc_train_base_bwd_f32(const float* x, int32_t ldx, const float* g, float* dx, int32_t lddx, int32_t rows, int32_t width, void* stream);
size_t fc_train_colsum_ws_bytes(int32_t cols, int32_t rows);
int fc_train_colsum_f32(const float* a, int32_t lda, int32_t cols, int32_t rows, float* out, int32_t accumulate, void* ws, size_t ws_bytes,
                        void* stream);

/* One EdgeConv level of the DGCNN embedder in TRAINING mode (models/pytorch_gcn.py:23-47, 81-99): y_ij = P[idx_ij] + Q[i] (the 1x1
 * conv of cat(f_j - f_i, f_i) split by linearity into two per-point products), BatchNorm with batch statistics over all (i, j),
 * LeakyReLU(slope) (0.2 in the DGCNN, 0 = the PAConv embedder's ReLU), max over the k neighbours -- forward and backward.  idx [rows, k] holds GLOBAL row indices (k <= 255); idx == NULL
 * with k == 1 and Q == NULL is BatchNorm1d + LeakyReLU on a [rows, C] matrix (conv5).  stats [3C] = mean | rstd | biased variance.
 * bwd: prep -> column sums of t1, t2 (fc_train_colsum_f32) = d beta, d gamma -> scatter (dQ; dP by float atomics into a zeroed buffer, or
 * dP = NULL and fc_train_edge_bwd_gather_f32 over the edges sorted by target: fixed summation order, bit-reproducible). */
size_t fc_train_edge_ws_bytes(int32_t rows, int32_t C);
int fc_train_edge_stats_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C, float eps,
                            float* stats, void* ws, size_t ws_bytes, void* stream);
int fc_train_edge_fwd_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                          const float* stats, const float* gamma, const float* beta, float slope, float* out, int32_t ldo, uint8_t* arg, void* stream);
int fc_train_edge_bwd_prep_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                               const float* stats, const float* gamma, const float* beta, float slope, const uint8_t* arg, const float* g, int32_t ldg,
                               float* t1, float* t2, int32_t ldt, int32_t rows_pad, void* stream);
/* Pooling of the global embedder (models/pytorch_gcn.py:178-182): out [B, >= 2 width] = [max over the scene's M points | mean],
 * arg [B, width] = arg-max point; bwd: dt [B*M, lddt]. */
int fc_train_pool_fwd_f32(const float* t, int32_t ldt, int32_t width, int32_t B, int32_t M, float* out, int32_t ldo, int32_t* arg, void* stream);
int fc_train_pool_bwd_f32(const float* g, int32_t ldg, const int32_t* arg, int32_t width, int32_t B, int32_t M, float* dt, int32_t lddt, void* stream);
int fc_train_edge_bwd_gather_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                                 const float* stats, const float* gamma, const uint8_t* arg, const float* t1, int32_t ldt, const float* dbeta,
                                 const float* dgamma, const int32_t* order, const int32_t* offsets, float* dP, int32_t lddp, void* stream);
int fc_train_edge_bwd_scatter_f32(const float* P, int32_t ldp, const float* Q, int32_t ldq, const int32_t* idx, int32_t rows, int32_t k, int32_t C,
                                  const float* stats, const float* gamma, const uint8_t* arg, const float* t1, int32_t ldt, const float* dbeta,
                                  const float* dgamma, float* dP, int32_t lddp, float* dQ, int32_t lddq, void* stream);

/* ---- PAConv context embedder, training primitives (models/scene_seg_PAConv/model/pointnet2/paconv.py:31-54, 107-153; util/paconv_util.py:52-56;
 * lib/pointops/src/grouping/grouping_cuda_kernel.cu:28-46; lib/pointops/src/interpolation/interpolation_cuda_kernel.cu:90-195).  Row-major fp32
 * matrices with explicit pitches; *_pad arguments: rows the OUTPUT buffers hold (pad rows / columns are zeroed).  Gather backwards take the edges
 * sorted by source row (order, offsets) and sum in that fixed order -- no atomics. */
int fc_op_paconv_knn_f32(const float* xyz, const float* qxyz, int32_t* out, int32_t B, int32_t n, int32_t m, int32_t k, void* stream);
int fc_train_paconv_group_f32(const float* xyz, const float* feat, int32_t ldf, int32_t C, const float* qxyz, const int32_t* nidx, float* E, int32_t ldE,
                              float* gdiff, int32_t B, int32_t n, int32_t m, int32_t K, void* stream);
int fc_train_softmax_fwd_f32(const float* x, int32_t ldx, int32_t width, int32_t rows, float* y, int32_t ldy, void* stream);
int fc_train_softmax_bwd_f32(const float* y, int32_t ldy, const float* dy, int32_t lddy, int32_t width, int32_t rows, int32_t rows_pad, float* dx,
                             int32_t lddx, void* stream);
int fc_train_assign_fwd_f32(const float* G, int32_t ldg, const float* S, int32_t lds, int32_t m, int32_t Cout, int32_t rows, int32_t rows_pad, float* out,
                            int32_t ldo, void* stream);
int fc_train_assign_bwd_f32(const float* G, int32_t ldg, const float* S, int32_t lds, const float* dout, int32_t lddo, int32_t m, int32_t Cout, int32_t rows,
                            int32_t rows_pad, float* dG, int32_t lddg, float* dS, int32_t ldds, void* stream);
int fc_train_centerdiff_fwd_f32(const float* x, int32_t ldx, int32_t C, int32_t K, int32_t groups, float* E, int32_t ldE, void* stream);
int fc_train_centerdiff_bwd_f32(const float* dE, int32_t ldE, int32_t C, int32_t K, int32_t groups, float* dx, int32_t lddx, void* stream);
int fc_train_rows_gather_bwd_f32(const float* dout, int32_t ldo, int32_t col0, int32_t C, const int32_t* order, const int32_t* offsets, const float* wts,
                                 int32_t div, int32_t n_src, int32_t n_src_pad, float* dsrc, int32_t lds, void* stream);
int fc_train_three_nn_f32(const float* uxyz, const float* kxyz, int32_t B, int32_t nu, int32_t mk, int32_t* idx, float* w, void* stream);
int fc_train_interp_fwd_f32(const float* Fk, int32_t ldfk, int32_t C, const int32_t* idx, const float* w, int32_t rows, int32_t rows_pad, float* out,
                            int32_t ldo, void* stream);

/* ---- optimiser step on flat gradient buffers (train.py:112-120: clip_grad_norm_ + Adam.step; flowcompare_amd/shard.py FlatAdam) */
size_t fc_train_sqnorm_ws_bytes(int64_t n);
int fc_train_sqnorm_f32(const float* g, int64_t n, double* out, int32_t slot, void* ws, size_t ws_bytes, void* stream);
int fc_train_adam_f32(float* const* params, const int64_t* offsets, const int32_t* chunk_tensor, const int64_t* chunk_off, int32_t n_chunks, const float* g,
                      float* m, float* v, const float* coef, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FCFLOW_H */
