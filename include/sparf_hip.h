/* sparf_hip.h -- C ABI of libsparf_hip.so, the MI355X (gfx950) NeRF renderer hot path.
 *
 * The reference (google-research/sparf) is pure PyTorch: it has NO FFI for this path.
 * Each entry point below therefore replaces a Python-level function of the reference
 * (file:line under /root/reference) rather than an existing binding; INTEGRATION.md
 * shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked HOST; tensors are dense fp32
 *     row-major unless a byte blob is stated; `stream` is a hipStream_t passed as void*;
 *   - return value: 0 = ok, non-zero = error code (never throws, never allocates device
 *     memory, keeps no mutable global state; all work is enqueued on `stream`);
 *   - a "pass" is one network (coarse or fine) evaluated on nrays*nsamp sample rows:
 *     ray setup -> fused MLP -> alpha compositing, and its backward;
 *   - rows = nrays * nsamp <= 2^27 per call (return code 4 above; slice larger batches).  The saved-activation
 *     and gradient areas are addressed per 32-row tile block and have no size limit of their own: a training
 *     pass is bounded by memory only (sparf_save_bytes + sparf_bwd_workspace_bytes, ~9 KB per row in bf16).
 *   - prec: 0 = bf16 MFMA operands / fp32 accumulate, 1 = fp32 MFMA (parity mode), 2 = bf16x3 (operands split into
 *     bf16 head + tail, three bf16 MFMAs per product on the forward / data-gradient chains; saved buffers hold the
 *     bf16 head plane, the weight gradient accumulates head products in fp32).  Measured at the 4096-ray BASELINE
 *     shapes against a float64 referee: outputs <= 2.8e-5 with metric depth; with inverse depth
 *     (samples at |p| ~ 1e8) rendered outputs 5e-5 ... 1.1e-4, per-sample values up to 1.2e-2 -- the Python mirror
 *     therefore routes the last samples of every ray of such passes through mode 1 (far rows, below); parameter gradients 7e-3 relative L2 under a random linear loss,
 *     2e-3 under the photometric loss (fp32: 1e-3 / 3e-4).
 */
#ifndef SPARF_HIP_H
#define SPARF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPARF_ABI_VERSION 6    /* 6: upstream gradients of EVERY output of a composite (depth_var, rgb_var, all_cumulated, density, rgb_samples) in
                                  sparf_pass_bwd_t / sparf_segment_t; stand-alone sparf_composite_forward / _backward; 5: SPARF_SAVE_Q8 on a pass's precision id; 4: far rows of a pass (the last K samples of every ray through a second precision); 3: ray segments of a pass (sparf_segment_t), tile-block save areas without a 2^31-byte limit,
                                  device-side Adam step counter; 2: band weights per pass, photometric-loss workspace */
#define SPARF_MAX_SEGMENTS 16
#define SPARF_PREC_BF16 0
#define SPARF_PREC_FP32 1
#define SPARF_PREC_X3 2        /* "bf16x3": bf16 MFMA on head + tail operands, three MFMAs per product; outputs within 1e-4 of fp32 at
                                  metric depth (see the note on inverse depth above) */
/* Save format of a training pass (ABI 5), OR-ed onto the `prec` of sparf_pass_forward / sparf_pass_backward / sparf_save_bytes /
 * sparf_bwd_workspace_bytes (every other entry point takes the plain precision id; weights and tables are those of the plain id):
 * SPARF_PREC_BF16 | SPARF_SAVE_Q8 or SPARF_PREC_X3 | SPARF_SAVE_Q8 keeps what the forward saves for the backward (layer inputs) and
 * what the data-gradient kernel hands the weight-gradient kernel (pre-activation gradients) as 8-bit integers on a linear grid
 * with one fp32 step per sample row and vector, x ~ (u - 128) * max|x| / 127, instead of bf16: half of the ~20 GB a 1M-row training
 * step moves through HBM.  The forward arithmetic -- every output of the pass -- and the data gradients (d_center, d_dir) are those
 * of the plain precision bit for bit; the WEIGHT gradients carry ~1.75x the rounding error of the bf16 saves (measured,
 * DESIGN.md 6).  Not combinable with far rows (far_count > 0). */
#define SPARF_SAVE_Q8 16
#define SPARF_N_LAYERS 10      /* mlp_feat.0..7, mlp_rgb.0..1 */
#define SPARF_N_PARAMS 530052  /* weights + biases of one network, flat (W0,b0,W1,b1,...) */

int sparf_abi_version(void);

/* ---- static layout tables (host) ------------------------------------------------------
 * int32 gather tables describing the packed weight streams; build once per precision on
 * the host, upload, and pass the device copy to sparf_pack_weights / sparf_pass_backward. */
int64_t sparf_table_count(int prec);                       /* number of int32 entries */
int sparf_build_tables(int prec, int32_t* host_out);       /* HOST pointer */

/* introspection of the packed weight streams (used by the layout tests): chunk `id` of the
 * forward (backward=0) or dgrad (backward=1) stream ->
 * out = {layer, segment, first m-block, m-blocks, first k-step, k-steps, byte offset, bytes} */
int sparf_stream_nchunks(int prec, int backward);
int sparf_stream_chunk(int prec, int backward, int id, int32_t out[8]);   /* HOST pointer */

/* ---- weight packing --------------------------------------------------------------------
 * Replaces the implicit use of nn.Linear weights by F.linear in
 * source/models/frequency_nerf.py:162-170, 215-219.
 * param_ptrs: HOST array of 20 device pointers {W0,b0,...,W9,b9} in nn.Linear layout
 * (W_l is [out][in] row-major: 256x63, 256x256 x3, 256x319, 256x256 x2, 257x256,
 * 128x283, 3x128).  Call again whenever a parameter changes. */
int64_t sparf_packed_bytes(int prec);
int sparf_pack_weights(int prec, const float* const* param_ptrs, const int32_t* tables, void* packed_out, void* stream);

/* BARF coarse-to-fine band weights of NeRF.positional_encoding (frequency_nerf.py:248-253):
 * out16 = {w_0..w_9 (points, L=10), w_0..w_3 (view, L=4), 0, 0} from the DEVICE scalar
 * `progress` (no host sync); all ones when has_c2f == 0 (opt.barf_c2f is None; progress may
 * then be NULL).  Not cached anywhere: trainers rewrite progress through `.data` every
 * iteration (nerf_trainer.py:273-275), so each pass is handed the vector computed for it and
 * its backward receives the same one. */
int sparf_c2f_weights(const float* progress, int has_c2f, float c2f_start, float c2f_end, float* out16, void* stream);

/* ---- depth sampling --------------------------------------------------------------------
 * sparf_sample_coarse replaces Graph.sample_depth (source/models/renderer.py:383-419) and
 * Graph.sample_depth_diff_max_range_per_ray (:595-624):
 *   t[r][i] = (u + i)/nsamp * scale + dmin, u = jitter[r][i] or u_const when jitter==NULL
 *   (0.5 for deterministic modes, 1.0 for the per-ray-max variant);
 *   scale = dmax_ray[r] - dmin when dmax_ray != NULL; inverse != 0 -> t = 1/(t + 1e-8).
 * sparf_sample_fine replaces Graph.sample_depth_from_pdf (:421-456) + cat + sort
 * (:334-336): u_mid[n_fine] are the mid-points of the (shared) sampling grid; writes the
 * sorted union [nrays][n_coarse+n_fine] to t_out and, if t_fine != NULL, the unsorted
 * resampled depths [nrays][n_fine].
 * range_dev (both): optional DEVICE pointer to {dmin, dmax} -- trainers hand the renderer
 * data_dict.depth_range[0], a device tensor (renderer.py:97-108); when non-NULL it replaces the
 * float arguments (scale = range_dev[1] - range_dev[0] in fp32, as torch computes it; with
 * dmax_ray only range_dev[0] is read), so no host readback is needed. */
int sparf_sample_coarse(const float* jitter, float u_const, const float* dmax_ray, const float* range_dev, float dmin, float scale,
                        int inverse, int nrays, int nsamp, float* t_out, void* stream);
int sparf_sample_fine(const float* weights, const float* t_coarse, const float* u_mid, const float* range_dev, float dmin, float dmax,
                      int nrays, int n_coarse, int n_fine, float* t_fine, float* t_out, void* stream);
/* (ABI 6) the same with the grid mid-points as a HOST array of n_fine <= 256 floats, copied into the launch arguments at the call:
 * the reference draws the grid on the CPU (renderer.py:439 `torch.rand(n_samples_fine+1)`), and a host -> device copy per render call
 * sits between the kernels of an otherwise copy-free step (round 5 timeline: every such copy was followed by 50-100 us of idle GPU). */
int sparf_sample_fine_hostgrid(const float* weights, const float* t_coarse, const float* u_mid_host, const float* range_dev, float dmin,
                               float dmax, int nrays, int n_coarse, int n_fine, float* t_fine, float* t_out, void* stream);

/* ---- ray generation ----------------------------------------------------
 * Replaces camera.get_center_and_ray / get_center_and_ray_at_pixels
 * (source/utils/camera.py:347-416; img2cam :296-306, cam2world :321-326) for the pixels a
 * render call actually uses (the reference builds all H*W rays of every image and indexes,
 * renderer.py:273-291):  g = K^-1 [x,y,1];  center = -R^T t;  ray = (R^T g + center) - center,
 * un-normalised.  Exactly one of pixels / ray_idx is non-NULL; flat indices address pixel
 * centres (+0.5, camera.py:365-368), explicit pixels are used as given (:400-406);
 * per_image != 0: one row of pixels / indices per image, else one row shared by all images.
 * sparf_ray_gen_backward: d_pose[nimg][3][4] from d_center / d_ray (either may be NULL);
 * intrinsics carry no gradient. */
int sparf_ray_gen_forward(const float* pose, const float* intr, const float* pixels, const int64_t* ray_idx, int per_image,
                          int width, int nimg, int nrays, float* center, float* ray, void* stream);
int sparf_ray_gen_backward(const float* pose, const float* intr, const float* pixels, const int64_t* ray_idx, int per_image,
                           int width, int nimg, int nrays, const float* d_center, const float* d_ray, float* d_pose,
                           void* stream);

/* ---- optimiser step ----------------------------------------------------
 * torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm) (source/training/base.py:96-97,
 * engine after_backward; skipped when max_norm <= 0) followed by torch.optim.Adam.step()
 * (source/training/nerf_trainer.py:181-185: betas, eps, no weight decay / amsgrad) for ONE
 * network: params = the 20 tensors W0,b0,...,W9,b9 (updated in place), grad = the flat
 * N_PARAMS gradient written by sparf_pass_backward, exp_avg / exp_avg_sq = flat Adam state,
 * step = 1-based update count, workspace = sparf_adam_workspace_floats() floats,
 * norm_out (optional) receives the gradient norm before clipping. */
int64_t sparf_adam_workspace_floats(void);
int sparf_adam_step(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace,
                    float* norm_out, float lr, float beta1, float beta2, float eps, int step, float max_norm, void* stream);
/* the same update with the step count kept on the DEVICE: *step_dev (int32, start at 0) is advanced by the call and
 * Adam's bias corrections are computed from it in the kernel.  For training steps captured in a hipGraph, whose
 * replays repeat the launch arguments verbatim (a host-side step number would freeze at its capture value). */
int sparf_adam_step_dev(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace,
                        float* norm_out, float lr, float beta1, float beta2, float eps, int* step_dev, float max_norm, void* stream);

/* BaseLoss.MSE_loss (kind 0) / huber_loss with delta (kind 1) of source/training/core/base_losses.py:151-156
 * on pred[n] and, if non-NULL, pred_fine[n] against target[n], summed as in base_losses.py:303-311;
 * writes the scalar loss and (where non-NULL) its derivatives w.r.t. pred / pred_fine.  workspace:
 * sparf_photometric_workspace_floats() floats (may be NULL for n <= 65536: single-workgroup path);
 * larger inputs (full images) are reduced by up to 256 workgroups in a fixed order. */
int64_t sparf_photometric_workspace_floats(void);
int sparf_photometric_loss(const float* pred, const float* pred_fine, const float* target, int64_t n, int kind, float delta,
                           float* loss, float* d_pred, float* d_pred_fine, float* workspace, void* stream);

/* ---- ray segments of a pass -------------------------------------------------
 * The SPARF losses issue 5-6 independent render calls per iteration (corres_loss.py:158-166,
 * depth_cons_loss.py:192,267,291).  Rays are independent, so several calls can share ONE pass: their rays
 * sit back to back in the pass's ray buffers (ray generation and depth sampling write each call's rows at
 * its offset) and a segment table tells the per-ray kernels what differs per call: the density-noise
 * scale (noise only on train-mode calls, frequency_nerf.py:191-192) and, in the backward, where each
 * call's upstream gradients live (autograd hands them over per call: no gather / scatter copies).
 * nseg == 0: the pass is one segment described by the pass-level fields.  Segments must tile
 * [0, nrays) in order.  HOST array, at most SPARF_MAX_SEGMENTS entries, copied at the call. */
typedef struct {
    int ray0, nrays;           /* rays [ray0, ray0 + nrays) of the pass */
    float noise_scale;         /* this segment's density-noise scale (0: no noise even if the pass has a noise tensor) */
    const float *g_rgb, *g_depth, *g_opacity, *g_weights;   /* backward only: THIS segment's upstream gradients
                                                               ([nrays][3], [nrays], [nrays], [nrays][nsamp]) or NULL */
    /* (ABI 6) ... and of the other outputs of the composite: [nrays] x 3, [nrays][nsamp], [nrays][nsamp][3], or NULL */
    const float *g_depth_var, *g_rgb_var, *g_all_cumulated, *g_density, *g_rgb_samples;
} sparf_segment_t;

/* ---- one network pass, forward ---------------------------------------------------------
 * Replaces NeRF.forward_samples + NeRF.composite
 * (source/models/frequency_nerf.py:260-281, 172-226, 283-343; camera.py:418-437). */
typedef struct {
    int prec, nrays, nsamp;
    const float* center;       /* [nrays][3] ray origins */
    const float* dir;          /* [nrays][3] ray directions, unnormalised */
    const float* t;            /* [nrays][nsamp] sample depths */
    const float* noise;        /* [nrays][nsamp] N(0,1) draws or NULL (frequency_nerf.py:191-192) */
    float noise_scale;         /* opt.nerf.density_noise_reg */
    int white_bg;              /* opt.nerf.setbg_opaque or opt.mask_img (frequency_nerf.py:337-338) */
    const void* packed;        /* sparf_pack_weights output for this network */
    const float* c2f;          /* [16] sparf_c2f_weights output for this pass */
    void* save;                /* sparf_save_bytes() bytes to keep for backward, or NULL (inference) */
    void* venc_ws;             /* scratch: nrays * 32 * (prec==0 ? 2 : 4) bytes */
    /* outputs */
    float* raylen;             /* [nrays] |dir| */
    float* sigma_raw;          /* [nrays][nsamp] density before noise/softplus */
    float* rgb_samples;        /* [nrays][nsamp][3] */
    float* density;            /* [nrays][nsamp] softplus(raw + noise) */
    float* weights;            /* [nrays][nsamp] */
    float* rgb;                /* [nrays][3] */
    float *depth, *opacity, *depth_var, *rgb_var, *all_cumulated;   /* [nrays] */
    int nseg;                  /* 0, or the number of ray segments */
    const sparf_segment_t* seg;   /* HOST array [nseg] (only ray0, nrays, noise_scale are read here) */
    /* far rows (ABI 4): far_count = K > 0 sends the LAST K samples of every ray through the kernels of far_prec as well -- a
     * second fused-MLP launch over nrays*K rows whose raw density / colour replace the main launch's for those samples before
     * compositing.  For inverse-depth sampling (source/models/renderer.py:413-416: sample i sits at t = 1/(1 - (u+i)/N + 1e-8),
     * the last one at t = N/(1-u), up to 1e8; after the merge of renderer.py:334-336 the last coarse samples are still the last
     * samples of the fine pass) under prec 2: the network is evaluated at |p| ~ t, a head + tail operand loses 8 bits against
     * fp32 there, and the samples at large t are where the rendered outputs miss 1e-4 (profiles/r04_inverse_routing_study.json:
     * all rows bf16x3 1.1e-4, last sample fp32 6.8e-5, last 8 samples fp32 1.2e-5).  0 < K < nsamp; far_prec must be 1 (fp32: the
     * kernels row routing is compiled into) and prec 0 or 2.  far_packed: the weights packed for far_prec; far_venc_ws: scratch of
     * nrays * 32 * 4 bytes; far_ws (training passes, save != NULL): scratch of sparf_save_bytes(far_prec, nrays*K) bytes, free
     * again when the call's work has run: what the far launch saved there -- activations and ReLU masks of the far rows -- is copied
     * into `save` at those rows (rounded to the main precision's bf16 plane), so that sparf_pass_backward, which knows nothing
     * about far rows, differentiates the forward that was composited: same operands as for every other row, the fp32 forward's
     * ReLU decisions. */
    int far_count, far_prec;
    const void* far_packed;
    void* far_ws;
    void* far_venc_ws;
    /* far_count = -1: far TILES by value, inference passes only (save == NULL), nsamp % 32 == 0, prec 2 only (the two launches must
     * cut the rows into the same 128-row workgroup tiles: prec 0 runs 256-row tiles and is refused), depth samples increasing along
     * every ray: each 128-row tile whose largest depth sample exceeds far_thr is evaluated by the far_prec kernel, every other tile
     * by the prec kernel -- for render_to_max passes (renderer.py:595-624: samples up to a per-ray far bound, so "the last K
     * samples" means nothing) under inverse depth: a ray is rendered up to a depth that is usually small, and only where it is
     * not does the pass need fp32. */
    float far_thr;
} sparf_pass_fwd_t;
int64_t sparf_save_bytes(int prec, int64_t rows);
int sparf_pass_forward(const sparf_pass_fwd_t* a, void* stream);

/* ---- one network pass, backward --------------------------------------------------------
 * Replaces torch.autograd through the functions above.  Upstream gradients may be NULL
 * (treated as zero).  grad_params receives d loss / d (W0,b0,...) flat, SPARF_N_PARAMS
 * floats.  d_center / d_dir (both or neither) request the gradients w.r.t. the rays
 * (joint pose optimisation); depth samples, `progress` and the noise receive none, as in
 * the reference (renderer.py:323 no_grad, frequency_nerf.py:251 .data). */
typedef struct {
    int prec, nrays, nsamp;
    const float *center, *dir, *t, *noise;
    float noise_scale;
    int white_bg;
    const void* packed;
    const float* c2f;          /* the vector the forward of this pass was given */
    const int32_t* tables;     /* device copy of sparf_build_tables output */
    const void* save;          /* written by sparf_pass_forward */
    const float *raylen, *sigma_raw, *rgb_samples, *weights;   /* forward outputs */
    const float *g_rgb, *g_depth, *g_opacity, *g_weights;      /* [nrays][3], [nrays], [nrays], [nrays][nsamp] */
    void* ws;                  /* sparf_bwd_workspace_bytes() bytes */
    float* grad_params;        /* [SPARF_N_PARAMS] */
    float *d_center, *d_dir;   /* [nrays][3] or NULL */
    int nseg;                  /* 0 (g_* above cover the whole pass), or the number of ray segments: then the */
    const sparf_segment_t* seg;   /* upstream gradients are read per segment from this HOST array [nseg] */
    /* (ABI 6) NeRF.composite is plain autograd in the reference: EVERY key it returns carries a gradient
     * (source/models/frequency_nerf.py:317-338).  Upstream gradients of depth_var / rgb_var / all_cumulated [nrays], of the
     * per-sample density [nrays][nsamp] (after softplus) and colour [nrays][nsamp][3] (after the sigmoid); any may be NULL. */
    const float *g_depth_var, *g_rgb_var, *g_all_cumulated, *g_density, *g_rgb_samples;
    /* (ABI 6) != 0: d_center / d_dir are ADDED to what the buffers hold (the fine pass of a render on top of its coarse pass: both
     * passes differentiate the same rays, source/models/renderer.py:304-343) */
    int accumulate_rays;
} sparf_pass_bwd_t;
int64_t sparf_bwd_workspace_bytes(int prec, int nrays, int nsamp, int pose);
int sparf_pass_backward(const sparf_pass_bwd_t* a, void* stream);

/* ---- stand-alone compositing (ABI 6) -----------------------------------------------------
 * NeRF.composite (source/models/frequency_nerf.py:283-343) as a function of caller-built per-sample values: `density`
 * [nrays][nsamp] (already through softplus), `rgb_samples` [nrays][nsamp][3] (already through the sigmoid), depth samples `t`
 * [nrays][nsamp], ray directions `dir` [nrays][3] (only their length enters: dist = delta * |dir|, :302-308).  The fused pass
 * above composites its own samples; this entry point serves callers that evaluate the network themselves (NeRF.forward /
 * forward_samples on explicit points) and composite afterwards.  Outputs as in sparf_pass_fwd_t; raylen [nrays] is written too.
 * Backward: upstream gradients of all seven outputs (any may be NULL) -> d_density [nrays][nsamp], d_rgb_samples [nrays][nsamp][3],
 * d_dir [nrays][3] (NULL to skip).  The depth samples receive no gradient (they carry none anywhere in the reference's callers:
 * renderer.py:323 no_grad, :405-407 fresh draws); the Python mirror refuses depth samples that require one. */
typedef struct {
    int nrays, nsamp, white_bg;
    const float *dir, *t, *density, *rgb_samples;
    float *raylen, *weights, *rgb, *depth, *opacity, *depth_var, *rgb_var, *all_cumulated;
} sparf_composite_fwd_t;
int sparf_composite_forward(const sparf_composite_fwd_t* a, void* stream);
typedef struct {
    int nrays, nsamp, white_bg;
    const float *dir, *t, *density, *rgb_samples, *raylen, *weights;       /* raylen, weights: written by the forward */
    const float *g_rgb, *g_depth, *g_opacity, *g_weights, *g_depth_var, *g_rgb_var, *g_all_cumulated;
    float *d_density, *d_rgb_samples;      /* [nrays][nsamp], [nrays][nsamp][3] */
    float *d_dir;                          /* [nrays][3] or NULL */
    float *d_len_ws;                       /* scratch [nrays] (needed when d_dir != NULL) */
} sparf_composite_bwd_t;
int sparf_composite_backward(const sparf_composite_bwd_t* a, void* stream);

/* ---- single-kernel entry points (measurement only) --------------------------------------
 * Launch exactly one of the three heavy kernels of a pass with the arguments the pass-level
 * calls would give it, so that bench.py can time each with stream events and rocprofv3 can
 * be cross-checked kernel by kernel.  which: 0 = fused MLP forward (saves activations iff
 * fwd->save != NULL), 1 = fused MLP dgrad, 2 = wgrad (+ its reduce kernel); 3 / 4 = the dgrad
 * kernel pinned to its 256-row (8 waves) / 128-row (4 waves) workgroup geometry -- bf16x3 has both
 * and sparf_pass_backward picks one per launch from the row count; other modes ignore the pin.
 * For 1 - 4 the workspace must already hold d_sigma / d_z (i.e. a full sparf_pass_backward ran). */
int sparf_launch_kernel(int which, const sparf_pass_fwd_t* fwd, const sparf_pass_bwd_t* bwd, void* stream);

/* ---- calibration (measurement only) -----------------------------------------------------
 * Two fixed kernels that know nothing of the renderer (csrc/calib.hip), for bench.py to time before and after its measurement:
 * the renderer's kernels change from round to round, these do not: they say what state (clocks under the power cap) a lease
 * found the chip in.  No reference counterpart (the reference has no device code).
 * sparf_calib_mfma: `iters` x 16 back-to-back v_mfma_f32_32x32x16_bf16 per wave on operands with random bits, two waves per SIMD on
 *   every CU; returns the bf16 flops the launch issues (> 0), or < 0 on bad arguments / launch failure.
 * sparf_calib_hbm: mode 0 = read [src, src + bytes) once through LDS-DMA (global_load_lds_dwordx4 nt, the weight-gradient kernel's
 *   operand path); mode 1 = copy it to dst (16-byte loads, non-temporal stores).  bytes: a multiple of 1024.
 * sink: SPARF_CALIB_SINK_FLOATS floats of device memory (never written in practice; keeps the kernels' results alive). */
#define SPARF_CALIB_SINK_FLOATS (1 << 18)
int64_t sparf_calib_mfma(int iters, float* sink, void* stream);
int sparf_calib_hbm(const void* src, void* dst, int64_t bytes, int mode, float* sink, void* stream);

/* Host arithmetic only (tests): the split-K decomposition sparf_pass_backward uses for the weight gradient of a pass of
 * rows_total sample rows whose active range (segments with an upstream gradient) covers rows_active rows.  nsplit_total is
 * what sparf_bwd_workspace_bytes reserved partial blocks for; nsplit_active <= nsplit_total always holds. */
int sparf_debug_wgrad_split(int64_t rows_total, int64_t rows_active, int* nsplit_total, int* nsplit_active, int* rows_per_split_active);

#ifdef __cplusplus
}
#endif
#endif
