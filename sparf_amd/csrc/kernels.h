// Internal launch interface between the C ABI (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sparf {

struct MlpFwdArgs {
    const char* packed;     // packed weight blob of this network (sparf_pack_weights)
    const float* c2f;       // [16] coarse-to-fine band weights of this pass (sparf_c2f_weights)
    const float* center;    // [nrays][3] ray origins
    const float* dir;       // [nrays][3] ray directions (unnormalised)
    const void* venc;       // [nrays][32] encoded view direction (act_t, pos layout)
    const float* t;         // [rows] sample depths, rows = nrays * nsamp
    int64_t rows;
    int nsamp;
    float* sigma_raw;       // [rows] raw density (before noise / softplus)
    float* rgb;             // [rows][3] colour after sigmoid
    void* save;             // saved activations [SAVE_COLS columns] or nullptr
    // Row routing (sparf_hip.h "far rows"): kernel row r stands for sample row g(r) = (r / nsamp) * row_stride + row_off + r % nsamp
    // of the pass: it reads its depth sample from t[g] and writes its outputs there; its ray is r / nsamp.  A whole pass:
    // stride = nsamp, offset 0 (g = r).  The far launch of a pass whose last K samples per ray go through another precision:
    // rows = nrays * K, nsamp = K, stride = the pass's samples per ray, offset = stride - K; its save area is its own.
    int row_stride = 0, row_off = 0;       // stride 0: no routing (g = r)
    // Tile routing by value (inference kernels only; sparf_hip.h far_count = -1): the workgroup evaluates a tile of its 32-row
    // wave tiles only if (max depth sample of the tile > tile_thr) == (tile_take == 2); tile_take 0: every tile.  The depth samples
    // of a ray are increasing and nsamp is a multiple of 32, so a wave tile's maximum is its last row.
    float tile_thr = 0.0f;
    int tile_take = 0;
};
// g(r) of the row-routing fields above.  32-bit arithmetic (a pass has at most 2^27 rows) behind a wave-uniform test, evaluated
// where it is needed instead of being kept live: the main launches (stride 0) pay nothing, and a 64-bit division would cost the
// register-bound fused kernels VGPRs they do not have.
static __host__ __device__ inline int64_t routed_row(int64_t r, int nsamp, int row_stride, int row_off) {
    if (row_stride == 0) return r;
    const unsigned ur = (unsigned)r, q = ur / (unsigned)nsamp;
    return (int64_t)(q * (unsigned)row_stride + (unsigned)row_off + (ur - q * (unsigned)nsamp));
}
enum { FWD_INFER = 0, FWD_SAVE_PLANES = 1, FWD_SAVE_Q8 = 2 };      // what the forward kernel leaves behind for the backward
int launch_mlp_fwd(int prec, int save, const MlpFwdArgs& a, int grid, hipStream_t stream);

struct MlpBwdArgs {
    const char* packed;
    const float* c2f;          // [16] band weights the forward of this pass used
    const float *center, *dir, *t;   // only read by the pose-gradient variant
    int64_t rows;              // end of the active row range (exclusive); whole pass: nrays * nsamp
    int nsamp;
    const void* save;          // activations saved by the forward kernel
    void* grad;                // [GRAD_COLS columns] pre-activation gradients (output)
    const float* d_sigma_raw;  // [rows]
    const float* d_z;          // [rows][3]
    float* dp;                 // [rows][3]  gradient w.r.t. the sample point (pose variant)
    float* dv;                 // [rows][32] gradient w.r.t. the encoded view dir (pose variant)
    int64_t row_begin;         // first active row (multiple of 32): rows before it belong to ray segments without upstream gradient
    int64_t rows_total;        // rows of the whole pass (= what the save / gradient areas were sized for)
};
// q8: the save / gradient areas are in the 8-bit format (layout.h AREA_Q8; bf16-operand modes)
// waves: workgroup geometry of the bf16x3 kernel, 8 (256-row tiles, the default) or 4 (128-row tiles); other precisions have one geometry
int launch_mlp_bwd(int prec, bool pose, bool q8, const MlpBwdArgs& a, int grid, hipStream_t stream, int waves = 8);

struct WgradArgs {
    const void* save;          // saved activations (X operands)
    const void* grad;          // pre-activation gradients (dY operands)
    int64_t rows;              // end of the active row range (exclusive)
    int rows_per_split;        // multiple of 32
    float* partial;            // [nsplit][wpartial_floats()]
    int64_t row_begin;         // first active row (multiple of 32)
};
// accumulate: grad_out += the reduced partials (the far launch of a routed pass, after the main one wrote grad_out)
int launch_wgrad(int prec, bool q8, const WgradArgs& a, int nsplit, const int32_t* wsrc, float* grad_out, hipStream_t s, bool accumulate = false);
// the two halves of launch_wgrad, for the chunked dgrad || wgrad schedule of sparf_pass_backward (api.hip)
int launch_wgrad_partials(int prec, bool q8, const WgradArgs& a, int nsplit, hipStream_t s);
int launch_wgrad_reduce(const float* partial, int nsplit, const int32_t* wsrc, float* grad_out, hipStream_t s, bool accumulate);

// ray segments of a pass (include/sparf_hip.h sparf_segment_t), by value in the kernel arguments.  Read with
// compile-time indices only (an unrolled select chain): a kernel-argument array indexed with a run-time value
// would be demoted to scratch memory.
enum { MAX_SEGMENTS = 16 };
struct SegTable {
    int n;                                     // 0: one segment = the pass-level fields
    int ray0[MAX_SEGMENTS];
    float noise_scale[MAX_SEGMENTS];
    const float *g_rgb[MAX_SEGMENTS], *g_depth[MAX_SEGMENTS], *g_opacity[MAX_SEGMENTS], *g_weights[MAX_SEGMENTS];
    // (ABI 6) upstream gradients of the other outputs of a composite: depth_var, rgb_var, all_cumulated [nrays], density [nrays][nsamp],
    // rgb_samples [nrays][nsamp][3]
    const float *g_depth_var[MAX_SEGMENTS], *g_rgb_var[MAX_SEGMENTS], *g_all_cum[MAX_SEGMENTS], *g_density[MAX_SEGMENTS], *g_rgb_samples[MAX_SEGMENTS];
};

struct CompositeFwdArgs {
    int nrays, nsamp;
    const float* t;            // [nrays][nsamp]
    const float* sigma_raw;    // [nrays][nsamp]
    const float* noise;        // [nrays][nsamp] or nullptr
    float noise_scale;
    const float* rgb_samples;  // [nrays][nsamp][3]
    const float* raylen;       // [nrays]
    int white_bg;
    float *weights, *density;                                           // [nrays][nsamp]
    float *rgb, *depth, *opacity, *depth_var, *rgb_var, *all_cumulated; // per ray ([nrays][3] for rgb)
    SegTable seg;
    int direct = 0;            // 1: `sigma_raw` holds the DENSITY itself (after softplus; no noise): the stand-alone composite of sparf_composite_forward
};
struct CompositeBwdArgs {
    int nrays, nsamp;
    const float *t, *sigma_raw, *noise;
    float noise_scale;
    const float *rgb_samples, *raylen, *weights;
    int white_bg;
    const float *g_rgb, *g_depth, *g_opacity, *g_weights;   // upstream gradients, any may be nullptr
    float* d_sigma_raw;        // [nrays][nsamp]
    float* d_z;                // [nrays][nsamp][3]  gradient before the colour sigmoid
    float* d_len;              // [nrays] gradient w.r.t. |ray| (nullptr to skip)
    SegTable seg;              // n > 0: upstream gradients per segment (g_* above unused)
    int ray_base;              // first ray of the launch (the active ray range of a segmented pass)
    // (ABI 6) the other outputs of NeRF.composite (frequency_nerf.py:317-338): any may be nullptr
    const float *g_depth_var = nullptr, *g_rgb_var = nullptr, *g_all_cum = nullptr;   // [nrays]
    const float* g_density = nullptr;       // [nrays][nsamp]
    const float* g_rgb_samples = nullptr;   // [nrays][nsamp][3]
    int direct = 0;            // 1: stand-alone composite -- `sigma_raw` is the density, d_sigma_raw receives d loss / d density, d_z d loss / d rgb_samples
};
struct RayGenArgs {
    int nimg, nrays, width, per_image;   // per_image: pixels / ray_idx have one row per image
    const float* pose;         // [nimg][3][4] world -> camera
    const float* intr;         // [nimg][3][3]
    const float* pixels;       // [(nimg)][nrays][2] (x, y) or nullptr
    const int64_t* ray_idx;    // [(nimg)][nrays] flat pixel indices or nullptr
    float* center;             // [nimg][nrays][3]
    float* ray;                // [nimg][nrays][3]
};
enum { SAMPLE_FINE_HOST_MAX = 256 };
struct SampleFineArgs {
    int nrays, n_coarse, n_fine;
    const float* weights;      // [nrays][n_coarse]
    const float* t_coarse;     // [nrays][n_coarse]
    const float* u_mid;        // [n_fine] interval mid-points of the (shared) sampling grid
    float dmin, dmax;
    const float* range_dev;    // {dmin, dmax} on the device (overrides the two floats) or nullptr
    float* t_fine;             // [nrays][n_fine] unsorted resampled depths (optional)
    float* t_out;              // [nrays][n_coarse+n_fine] sorted union
    // u_mid == nullptr: the grid mid-points travel BY VALUE in the kernel arguments (sparf_sample_fine_hostgrid: the reference draws
    // the grid on the CPU, renderer.py:439 -- handing it over as launch arguments costs no host -> device copy between the kernels)
    float u_host[SAMPLE_FINE_HOST_MAX];
};
struct RayReduceArgs {
    int nrays, nsamp;          // nrays = launch size, starting at ray_base
    const float *t, *dp, *dv, *dir, *raylen, *d_len, *c2f_view;
    float *d_center, *d_dir;
    int ray_base;
    int accumulate = 0;        // != 0: add to d_center / d_dir instead of overwriting them
};
// far rows of a pass: copy what the far (fp32) forward saved into the main (bf16-plane) save area, at the rows it stands for (ray_ops.hip)
int launch_far_transplant(int main_prec, const void* far_area, void* main_area, int64_t frows, int far_count, int nsamp, hipStream_t s);
// d_dir = d_len * dir / |dir| (the stand-alone composite's only dependence on the ray: dist = delta * |ray|, frequency_nerf.py:302-308)
int launch_len_to_dir(const float* dir, const float* raylen, const float* d_len, int nrays, float* d_dir, hipStream_t s);
// prec < 0: raylen only (venc / c2f_view may be nullptr)
int launch_ray_setup(int prec, const float* dir, int nrays, const float* c2f_view, void* venc, float* raylen, hipStream_t s);
int launch_sample_coarse(const float* jitter, float u_const, const float* dmax_ray, const float* range_dev, float dmin, float scale,
                         int inverse, int64_t rows, int nsamp, float* t, hipStream_t s);
int launch_composite_fwd(const CompositeFwdArgs& a, hipStream_t s);
int launch_composite_bwd(const CompositeBwdArgs& a, hipStream_t s);
int launch_sample_fine(const SampleFineArgs& a, hipStream_t s);
int launch_ray_reduce(const RayReduceArgs& a, hipStream_t s);
int launch_adam(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace, float* norm_out,
                float lr, float beta1, float beta2, float eps, int step, int* step_dev, float max_norm, hipStream_t s);
int photometric_workspace_floats();
int launch_photometric_loss(const float* pred, const float* pred_fine, const float* target, int64_t n, int kind, float delta,
                            float* loss, float* d_pred, float* d_pred_fine, float* workspace, hipStream_t s);
int launch_ray_gen_fwd(const RayGenArgs& a, hipStream_t s);
int launch_ray_gen_bwd(const RayGenArgs& a, const float* d_center, const float* d_ray, float* d_pose, hipStream_t s);

}  // namespace sparf
