/*
 * dad.h — C ABI of libdad_hip.so: the MI355X (gfx950) reverse-diffusion planning sampler.
 *
 * The reference (darshangm/dynamics-aware-diffusion) is 100 % Python and has no FFI seam;
 * its seam is the Python class API of m_diffuser.models / m_diffuser.guides that
 * scripts/evaluate.py consumes.  This library sits UNDER a Python
 * mirror of that API (dynamics_aware_diffusion_amd/) and is bound with ctypes.  Each entry
 * point names the reference function it replaces (paths under /root/reference/).
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success or a negative DAD_E_* code;
 *    dad_last_error() returns a thread-local message.  Nothing throws or aborts.
 *  - device pointers are fp32, contiguous, owned by the caller (torch-ROCm tensors);
 *    trajectories use the reference's external layout (batch, horizon, transition_dim).
 *  - all work is enqueued asynchronously on the caller's hipStream_t (passed as void*);
 *    step functions never allocate: the caller passes a workspace of dad_workspace_bytes().
 *  - no process-wide mutable state: every knob (precision, debug hooks, caches, graphs) lives on
 *    the dad_model; the only statics are the immutable kernel table and a mutex-guarded set of
 *    devices whose kernels had their LDS limit raised.  One dad_model per device; calls on
 *    different models may run on different threads (thread-compatible: one thread per model).
 *  - ONE STREAM PER MODEL AT A TIME: the split-K arrival tickets, the Philox key cell and the
 *    persistent-step flags are owned by the dad_model, so two calls on the same model must not be in
 *    flight on different streams concurrently (enqueue them on one stream, or synchronise between
 *    streams; concurrent loops need one dad_model each, as bench.py --inflight builds them).
 */
#ifndef DAD_H
#define DAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAD_OK 0
#define DAD_E_INVALID (-1)   /* bad argument / unsupported shape                      */
#define DAD_E_STATE (-2)     /* call order violated (e.g. step before finalize)       */
#define DAD_E_KEY (-3)       /* unknown or duplicate weight key, or shape mismatch    */
#define DAD_E_HIP (-4)       /* a HIP runtime call failed (message has hipGetErrorString) */
#define DAD_E_RANGE (-5)     /* timestep outside the loaded schedule (reference: RuntimeError
                                from gather, m_diffuser/models/diffusion.py:28)       */
#define DAD_E_WORKSPACE (-6) /* workspace too small                                   */

#define DAD_MAX_LEVELS 8

/* How the conv GEMMs multiply (dad_model_set_precision).  Both accumulate in fp32 and meet the same
 * fp32 parity tolerance against the reference (m_diffuser runs F.conv1d in fp32):
 *   FP32   exact fp32 products on v_mfma_f32_32x32x2_f32 (default);
 *   F16X3  every fp32 operand carried as two halves (hi + lo*2^-11, 22 significant bits), three
 *          v_mfma_f32_32x32x16_f16 per product block; activations must stay within +-65504. */
#define DAD_PREC_FP32 0
#define DAD_PREC_F16X3 1

typedef struct dad_model dad_model;
typedef void* dad_stream_t; /* hipStream_t */

/* Architecture of the denoiser + diffusion constants.
 * Mirrors TemporalUnet.__init__ (m_diffuser/models/temporal_unet.py:135-197) and
 * GaussianDiffusion.__init__ (m_diffuser/models/diffusion.py:62-94). */
typedef struct dad_cfg {
    int32_t transition_dim;           /* observation_dim + action_dim                     */
    int32_t dim;                      /* width of the sinusoidal embedding                */
    int32_t time_dim;                 /* time-embedding width (reference: == dim)         */
    int32_t n_levels;                 /* len(dim_mults)                                   */
    int32_t channels[DAD_MAX_LEVELS]; /* dim * dim_mults[i] per level                     */
    int32_t kernel_size;              /* 3, 5 or 7 (temporal_unet.py:139; 5 in every recipe)  */
    int32_t horizon;                  /* planning horizon H: a power of two with
                                         H / 2^(n_levels-1) >= 4 (the deepest level keeps at
                                         least 4 positions; else DAD_E_INVALID)              */
    int32_t n_timesteps;              /* length T of the trained schedule                 */
    int32_t predict_epsilon;          /* diffusion.py:192-197                             */
    int32_t clip_denoised;            /* diffusion.py:199-200                             */
} dad_cfg;

const char* dad_last_error(void);
const char* dad_version(void);

/* Replaces: TemporalUnet(...) / GaussianDiffusion(...) construction + .to(device). */
int dad_model_create(const dad_cfg* cfg, dad_model** out);
void dad_model_destroy(dad_model* m);

/* Replaces: load_state_dict (scripts/evaluate.py:198).  `key` is the reference's
 * state_dict key WITHOUT the leading "model." (SURVEY.md Appendix C); `data` is a HOST
 * fp32 pointer in the reference's layout; the library packs and uploads its own copy.
 * May be called again for the same key before the next finalize (weights updated). */
int dad_model_load_weight(dad_model* m, const char* key, const float* data,
                          const int64_t* shape, int32_t ndim);

/* The five schedule buffers the reverse step reads (diffusion.py:117-128), HOST fp32,
 * each of length cfg.n_timesteps: sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1, posterior_mean_coef2, posterior_log_variance_clipped. */
int dad_model_load_schedule(dad_model* m, const float* sqrt_recip, const float* sqrt_recipm1,
                            const float* coef1, const float* coef2, const float* log_var);

/* Optional.  The SinusoidalPosEmb table (temporal_unet.py:19-32) for t = 0 .. n_timesteps-1,
 * HOST fp32 (n_timesteps, dim), computed by the caller: the Python mirror evaluates the
 * reference's own torch expression so the table is bit-identical to what the reference computes
 * on the same host.  Without it dad_model_finalize evaluates the same formula with libm. */
int dad_model_load_time_embedding(dad_model* m, const float* emb, int32_t n_timesteps, int32_t dim);

/* No reference counterpart (the reference computes in whatever dtype the module holds, fp32 in
 * scripts/evaluate.py): selects the conv arithmetic, DAD_PREC_*.  Takes effect at the next
 * dad_model_finalize (weights are re-packed); a finalized model must be finalized again. */
int dad_model_set_precision(dad_model* m, int32_t precision);

/* Widths the reference accepts and the conv-GEMM tiles do not — GroupNorm(8, C) needs only C % 8 == 0
 * (temporal_unet.py:71: `--dim 48`, `--dim 96`), the tiles a multiple of 32 with a power-of-two C / 8: the host
 * language pads every GroupNorm group of such a level with zero channels up to the next power of two >= 4 (weights,
 * biases, gamma / beta of the padding are zero: the padded net computes the same function), gives the PADDED widths
 * in dad_cfg and states the real ones here; the GroupNorm statistics then count the real channels only.  Call
 * between dad_model_create and dad_model_finalize.  Such models run the batch kernels at every batch size; they
 * train as the padded net (dad_unet_backward fills padded gradient tensors: the entries of the padding are garbage
 * the host language drops; the GroupNorm backward counts the real channels). */
int dad_model_set_group_channels(dad_model* m, const int32_t* real_channels, int32_t n_levels);

/* Horizons the reference accepts and the tiles do not: its U-Net takes any length every level can halve
 * (temporal_unet.py:35-54: H % 2^(levels-1) == 0 — 24, 48, 96, 100 ...), the conv-GEMM tiles whole power-of-two
 * samples.  Give the next power of two in dad_cfg.horizon and the real horizon here: every activation keeps the padded
 * layout with ZERO rows behind the real ones (exactly the zero padding a conv sees at the end of a sample), the
 * GroupNorm statistics count the real rows only, and the external tensors (x, noise, guide gradient, outputs) keep
 * the real shape (B, real_horizon, transition_dim).  Call between dad_model_create and dad_model_finalize.  Such models
 * run the batch kernels at every batch size; they train (the data gradients are zero-padded the same way). */
int dad_model_set_horizon(dad_model* m, int32_t real_horizon);
/* Checks every tensor is present, builds the per-timestep time-embedding tables
 * (SinusoidalPosEmb + time_mlp + every block's Mish->Linear, temporal_unet.py:19-32,
 * 97-100,155-160 — batch-invariant during sampling) and the launch plan. */
int dad_model_finalize(dad_model* m, dad_stream_t stream);

/* Bytes of device scratch one call at batch size B needs: the activation buffers of the launch
 * plan plus, for batches small enough to use grid-level split-K, the partial-tile slabs. */
int dad_workspace_bytes(const dad_model* m, int32_t batch, size_t* bytes);

/* Replaces: TemporalUnet.forward(x, t) with one shared timestep t
 * (temporal_unet.py:199-241).  x, out: (B, H, td) device fp32.  out = eps_theta(x, t). */
int dad_unet_forward(dad_model* m, const float* x, int32_t t, float* out, int32_t batch,
                     void* workspace, size_t workspace_bytes, dad_stream_t stream);

/* Replaces: TemporalUnet.forward(x, t) with one timestep PER ROW, as the training objective calls
 * it (GaussianDiffusion.loss, m_diffuser/models/diffusion.py:253-290: t ~ randint per trajectory).
 * t_rows: (B) int32 on the DEVICE, every entry in [0, n_timesteps) (the caller checks the range:
 * the library cannot without a device synchronisation).  Inference form (time embeddings from the
 * per-timestep tables); the differentiable form is dad_unet_forward_train below. */
int dad_unet_forward_rows(dad_model* m, const float* x, const int32_t* t_rows, float* out, int32_t batch,
                          void* workspace, size_t workspace_bytes, dad_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training side: replaces loss.backward() through TemporalUnet in the reference's training step
 * (m_diffuser/utils/training.py:144-156; the objective is GaussianDiffusion.loss,
 * m_diffuser/models/diffusion.py:253-290).  The reference relies on torch autograd; here the
 * denoiser's backward pass is explicit:
 *   - data gradients of Conv1d / ConvTranspose1d (temporal_unet.py:35-54,70) run on the forward
 *     conv-GEMM kernels with transposed, tap-flipped weight images packed at dad_model_finalize;
 *   - weight gradients are an MFMA GEMM over the batch rows (csrc/train_bwd.hpp, conv_wgrad);
 *   - GroupNorm(8) + Mish backward (temporal_unet.py:71-72) is one kernel per conv, fed by the
 *     pre-normalisation output and the (mean, rstd) pairs the training forward keeps.
 * The time MLPs (SinusoidalPosEmb -> Linear -> Mish -> Linear and every block's Mish -> Linear,
 * temporal_unet.py:97-100,155-160) stay with the caller: the forward takes their per-row outputs
 * (B, temb_width) — the concatenation, in launch order, of every ResidualTemporalBlock's projection —
 * and the backward returns the gradient with respect to them.  fp32 only; no optimiser, no EMA.
 *
 * dad_model_set_training(m, 1) must precede dad_model_finalize (the extra weight images are packed
 * there).  dad_train_grad_info enumerates the gradient tensors (reference state_dict key without "model.",
 * element count; `offset` is their position in a packed buffer for callers that want one) — every tensor in
 * the reference's own layout (Conv1d (out, in, k); ConvTranspose1d (in, out, k)).  The time-MLP tensors are
 * not in the list. */
int dad_model_set_training(dad_model* m, int32_t on);
/* Replaces: nothing in the reference (its modules ARE the parameters); here the engine holds packed copies,
 * and an optimiser step (utils/training.py:166) changes the parameters every iteration.  Re-derives, ON THE
 * DEVICE, everything the engine keeps of the given tensors — packed conv images (forward and, in training
 * mode, the data-gradient images), biases, GroupNorm affine parameters, the per-timestep time tables — from
 * DEVICE fp32 tensors in the reference's layouts (same keys as dad_model_load_weight).  A conv whose block's
 * 1x1 residual conv rides in its image needs both weights in the same call.  fp32 arithmetic only.
 * Asynchronous on `stream`; captured loops stay valid (the images are updated in place). */
int dad_model_refresh_weights(dad_model* m, int32_t n, const char* const* keys, const float* const* tensors,
                              dad_stream_t stream);
int dad_train_grad_count(const dad_model* m, int32_t* count, int64_t* total_floats);
int dad_train_grad_info(const dad_model* m, int32_t i, const char** key, int64_t* offset, int64_t* numel);
/* saved: every activation of one forward (nothing is overwritten before the backward pass reads it);
 * scratch: gradient tensors and reduction slabs of one backward pass. */
int dad_train_workspace_bytes(const dad_model* m, int32_t batch, size_t* saved_bytes, size_t* scratch_bytes);
/* Replaces: TemporalUnet.forward(x, t) inside GaussianDiffusion.loss (diffusion.py:272) in training mode.
 * row_index: (B) int32 device, row b of temb_rows that sample b uses (normally 0..B-1); temb_rows:
 * (rows, temb_width) device fp32.  out = eps_theta (B, H, td); `saved` is handed to dad_unet_backward. */
int dad_unet_forward_train(dad_model* m, const float* x, const int32_t* row_index, const float* temb_rows,
                           float* out, int32_t batch, void* saved, size_t saved_bytes, dad_stream_t stream);
/* Replaces: autograd's walk from d loss / d eps back through the denoiser.  d_out: (B, H, td);
 * d_x (optional): (B, H, td) gradient w.r.t. the noisy trajectory; d_temb_rows: (B, temb_width), fully
 * overwritten; grad_tensors: one device tensor per entry of dad_train_grad_info, in that order, each of that
 * entry's element count and fully overwritten (separate tensors so that autograd can adopt them as .grad
 * without a copy; each call computes the gradient of ONE batch: accumulation over micro-batches is the
 * caller's add). */
int dad_unet_backward(dad_model* m, const float* x, const float* d_out, float* d_x, float* d_temb_rows,
                      float* const* grad_tensors, int32_t n_grad_tensors, int32_t batch, void* saved,
                      size_t saved_bytes, void* scratch, size_t scratch_bytes, dad_stream_t stream);

/* Arguments of one reverse step beyond (x, t). All pointers may be NULL unless noted. */
typedef struct dad_step_args {
    const float* noise;      /* (B,H,td) injected z; NULL => in-kernel Philox            */
    uint64_t seed;           /* Philox key (used when noise == NULL)                     */
    uint64_t row_offset;     /* global index of batch row 0 (multi-GPU shards)           */
    uint64_t draw;           /* Philox draw id of this step (loop uses T - t)            */
    const float* cond0;      /* inpainting value for horizon step 0: (td) or (B,td)      */
    int32_t cond_per_row;    /* 0: cond0 is (td) broadcast; 1: (B,td)                    */
    const float* guide_grad; /* (B,H,td) d guide/d x_t, or NULL                          */
    float guide_weight;      /* policies.py:97                                           */
    float* mean_out;         /* optional (B,H,td): posterior mean incl. guidance         */
    float* eps_out;          /* optional (B,H,td): raw model output                      */
} dad_step_args;

/* Replaces: GaussianDiffusion.p_sample (diffusion.py:205-223) and
 * GuidedPolicy.p_sample_with_guidance (guides/policies.py:65-112): U-Net, x0 prediction,
 * clamp, posterior mean, guidance nudge, noise, inpainting — x updated IN PLACE.
 * With args->mean_out set and x_out_disabled != 0 it is p_mean_variance only
 * (diffusion.py:182-203) and x is left untouched. */
int dad_denoise_step(dad_model* m, float* x, int32_t t, int32_t batch, const dad_step_args* args,
                     int32_t x_out_disabled, void* workspace, size_t workspace_bytes,
                     dad_stream_t stream);

/* Replaces: GaussianDiffusion.p_sample_loop (diffusion.py:225-251) and
 * GuidedPolicy.sample_loop without a guide (guides/policies.py:114-149): runs
 * t = n_steps-1 .. 0 on x (which must already hold x_T, with conditions applied).
 * noise_stack: (n_steps,B,H,td) injected z in loop order, or NULL for in-kernel Philox
 * (draw id of iteration j is j+1; draw 0 is reserved for x_T, see dad_fill_normal).
 * proj: optional projection applied after every step (README semantics; the shipped
 * reference never calls it — SURVEY.md F5), alphas[n_steps] indexed by t on the HOST.
 * use_graph != 0 replays a cached hipGraph of the whole loop: every pointer argument and
 * (n_steps, batch, row_offset) are frozen in the capture, so callers keep them stable (a new
 * combination is captured once; at most 16 graphs are cached).  The Philox seed is NOT frozen:
 * it is written to device memory ahead of each replay. */
typedef struct dad_project_args {
    const float* P;         /* (D,D) device, D = (H+1)*n + H*m, row-major             */
    const float* obs_mean;  /* device (od) */
    const float* obs_std;
    const float* act_mean;  /* device (ad) */
    const float* act_std;
    int32_t state_dim;      /* n */
    int32_t observation_dim;/* od (== n in every reachable reference configuration)  */
    int32_t action_dim;     /* m */
    /* Optional device scratch of at least batch * H * (od + m) floats, owned by the caller.  With it,
     * batches of 32+ trajectories run v @ P as an MFMA GEMM that reads P once per 32 trajectories
     * (required beyond D = 2000, where one trajectory's partial sums no longer fit a CU's LDS);
     * without it every trajectory streams P on its own. */
    float* scratch;
    size_t scratch_bytes;
} dad_project_args;

int dad_sample_loop(dad_model* m, float* x, int32_t n_steps, int32_t batch,
                    const float* noise_stack, uint64_t seed, uint64_t row_offset,
                    const float* cond0, int32_t cond_per_row,
                    const dad_project_args* proj, const float* proj_alphas_host,
                    int32_t use_graph, void* workspace, size_t workspace_bytes,
                    dad_stream_t stream);

/* Replaces: DynamicsAwarePolicy.apply_projection (guides/policies.py:409-485) for one
 * alpha (policies.py:358-383 is evaluated by the caller).  x: (B,H,od+m) IN PLACE. */
int dad_project(const dad_project_args* p, float alpha, float* x, int32_t batch,
                int32_t horizon, dad_stream_t stream);

/* Replaces: the arithmetic of ProjectionLoss.compute (m_diffuser/losses/__init__.py:161-186):
 * violation[b] = sum_d (v_b - v_b P)_d^2 with v the de-normalised concatenated trajectory
 * [s_0..s_{H-1}, s_{H-1}, a_0..a_{H-1}] — the caller divides the sum over rows by B * D.
 * x (B,H,od+m) is read only; violation: (B) device fp32. */
int dad_projection_violation(const dad_project_args* p, const float* x, float* violation, int32_t batch,
                             int32_t horizon, dad_stream_t stream);

/* Replaces: torch.randn(shape) for x_T (diffusion.py:241; policies.py:134) with the
 * library's counter-based generator: element e of global row r gets Philox4x32-10
 * (key = seed, counter = (draw, r*H*td + e)) -> Box-Muller.  Result is independent of how
 * rows are sharded over GPUs. */
int dad_fill_normal(float* x, int32_t batch, int32_t row_elems, uint64_t seed,
                    uint64_t row_offset, uint64_t draw, dad_stream_t stream);

/* Optional kernel timing: when enabled, the run of conv-GEMM launches of every denoiser
 * evaluation is bracketed by one pair of HIP events on the launch stream;
 * dad_profile_read synchronises, returns the summed duration, the number of conv-GEMM
 * launches inside the brackets and their algorithmic FLOPs, and resets the counters. */
int dad_profile_enable(dad_model* m, int32_t on);
int dad_profile_read(dad_model* m, double* conv_ms, int64_t* conv_launches, double* conv_flops);

/* Test / tuning hooks, all per model (two models in one process do not interact).
 * dad_debug_set_tile: force conv tile configuration `cfg` (0..9, see kTiles in csrc/host_plan.hpp)
 * wherever it is valid for a layer; -1 restores the heuristic; 100+cfg (99 = heuristic tile)
 * additionally disables grid-level split-K.
 * dad_debug_set_option: "fuse_residual" (the 1x1 residual conv rides in its block's first conv
 * launch; 0 = always its own launch), "xswz" (LDS slot shifts), "xcd_order" (XCD-aware tile
 * order), "split_target" (blocks a split-K layer aims for), "cc" (small batches take the
 * consumer-combine kernels of csrc/conv_cc.hpp; 0 = always the batch-256 kernels), "cc_max_rows"
 * (largest batch * horizon that does), "ccw_max_rows" (the same bound for nets whose small-batch plan
 * needs the streamed-weight kernels of csrc/conv_ccw.hpp: GroupNorm groups wider than 64 channels),
 * "ccw_min_blocks" (blocks a wide layer keeps when its K slices are fattened), "ccw_prefer16" (two
 * 16-row tiles instead of a 32-row tile whose K slice LDS would halve), "chain" (nets of dim <= 128 at
 * horizon 32: the five level-0 encoder launches as one launch per sample, csrc/conv_chain.hpp; off by
 * default — measured slower than the launches it replaces), "chain_min_batch", "wgrad_blocks" (blocks a
 * weight-gradient launch of the backward pass aims for: tiles x batch splits).
 * Results do not depend on these choices beyond fp32 summation order. */
int dad_debug_set_tile(dad_model* m, int32_t cfg);
int dad_debug_set_option(dad_model* m, const char* name, int32_t value);

/* Test hooks for direct comparison against the reference's intermediates (synchronous).
 * dad_debug_read_table copies row t of a per-timestep table to HOST memory (`capacity` floats
 * available; the row width is returned in *width_out):
 *   DAD_TABLE_SINUSOID  SinusoidalPosEmb(t)                      (dim floats;  temporal_unet.py:19-32)
 *   DAD_TABLE_TIME_MLP  time_mlp(t) = Linear(Mish(Linear(emb)))  (time_dim;    temporal_unet.py:155-160)
 *   DAD_TABLE_BLOCKS    every block's Linear(Mish(time_mlp(t)))  (sum of C_out over residual
 *                       blocks, in launch order;                  temporal_unet.py:97-100)
 * dad_debug_mish applies the conv epilogue's Mish to n device floats. */
#define DAD_TABLE_SINUSOID 0
#define DAD_TABLE_TIME_MLP 1
#define DAD_TABLE_BLOCKS 2
int dad_debug_read_table(dad_model* m, int32_t which, int32_t t, float* host_out, int32_t capacity,
                         int32_t* width_out);
int dad_debug_mish(const float* in, float* out, int64_t n, dad_stream_t stream);
/* 1 when the planner's statement of which conv-GEMM kernels exist equals the kernel registry (no device call) */
int dad_debug_kernel_table_consistent(void);
/* Which kernels a batch takes (host-side query, no device work): launches_out = conv launches of one
 * denoiser evaluation through the small-batch consumer-combine kernels (csrc/conv_cc.hpp), 0 when the
 * batch runs the batch-256 kernels; wide_out = how many of them are the streamed-weight form for
 * wide layers (csrc/conv_ccw.hpp). */
int dad_debug_small_batch_plan(dad_model* m, int32_t batch, int32_t* launches_out, int32_t* wide_out);

#ifdef __cplusplus
}
#endif
#endif /* DAD_H */
