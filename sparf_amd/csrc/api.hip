// C ABI of libsparf_hip.so (see include/sparf_hip.h): argument checking, workspace
// carving and kernel sequencing on the caller's stream.  No allocation, no global state.
#include "../../include/sparf_hip.h"

#include "kernels.h"
#include "streams.h"

namespace sparf {
int build_tables(int prec, int32_t* out);
int launch_pack(int prec, const float* const* param_ptrs_host, const int32_t* tables, void* out, hipStream_t s);
int launch_c2f(const float* progress, int has_c2f, float c2f_start, float c2f_end, float* out, hipStream_t s);
int launch_calib_mfma(int iters, float* sink, int grid, hipStream_t s);
int launch_calib_hbm(const void* src, void* dst, int64_t bytes, int mode, float* sink, int grid, hipStream_t s);
int64_t calib_mfma_flops(int iters, int grid);
}  // namespace sparf

using namespace sparf;

// layout constants, evaluated at compile time (as plain calls the constexpr functions of
// streams.h would re-run their enumeration loops on the host at every API call)
static constexpr int64_t kWsrcOff[N_PREC] = {tbl_wsrc_off(PREC_BF16), tbl_wsrc_off(PREC_FP32), tbl_wsrc_off(PREC_X3)};
static constexpr int64_t kPackedBytes[N_PREC] = {packed_bytes(PREC_BF16), packed_bytes(PREC_FP32), packed_bytes(PREC_X3)};
static constexpr int64_t kTblCount[N_PREC] = {tbl_count(PREC_BF16), tbl_count(PREC_FP32), tbl_count(PREC_X3)};
static constexpr int64_t kPartialFloats = wpartial_floats();

static inline bool prec_ok(int p) { return p >= 0 && p < N_PREC; }
// A pass's precision id may carry SPARF_SAVE_Q8 (sparf_hip.h): the arithmetic of `base`, save and gradient areas in the 8-bit format
// (layout.h AREA_Q8); bf16-operand modes only
struct PassPrec { int base; bool q8; int af; bool ok; };
static inline PassPrec pass_prec(int p) {
    PassPrec r;
    r.base = p & ~SPARF_SAVE_Q8;
    r.q8 = (p & SPARF_SAVE_Q8) != 0;
    r.ok = prec_ok(r.base) && (!r.q8 || r.base == PREC_BF16 || r.base == PREC_X3);
    r.af = area_format(r.ok ? r.base : 0, r.q8);
    return r;
}
static inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }
static inline int num_cus() {
    // CU count of the current device, looked up once per device: an immutable hardware attribute (the only
    // process-level state of the library), so that the pass calls issue no device queries -- they may be
    // running under hipGraph stream capture
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 256;
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cached[dev] = n > 0 ? n : 256;
    }
    return cached[dev];
}
static inline int mlp_grid(int prec, int64_t rows) {
    const int tile = nwaves_of(prec) * 32;
    const int64_t ntiles = (rows + tile - 1) / tile;
    const int cus = num_cus();
    return (int)(ntiles < cus ? ntiles : cus);
}
// Workgroup geometry of the bf16x3 data-gradient kernel for a launch over `rows` rows (mlp_dev.h PolicyX3DgradT): both kernels run
// one workgroup per CU striding over their tiles, so a launch lasts (rounds of tiles) x (time of one tile).  The 8-wave kernel's
// 256-row tile is 12-17 % cheaper per row, but its last round may be mostly empty: 32 768 rows (the coarse pass of a 512-ray
// step) are 128 of its tiles -- half the chip idle for a whole tile time -- and exactly one round of 128-row tiles.  A 128-row
// tile of the 4-wave kernel takes X3_W4_TILE_PCT % of a 256-row tile's time -- and a launch whose rows are some full rounds plus a poor
// last one can run the full rounds in 8 waves and the remainder in 4 (x3_dgrad_rows8 below) -- (measured, profiles/r06_dgrad_geometry.log: 0.066-0.071 ms
// against 0.097-0.112 ms per round; 32 768 rows 0.112 -> 0.071 ms, with pose gradients 0.115 -> 0.080; 98 304 rows a wash, every
// other shape of the table the 8-wave kernel by 5-12 %).
enum { X3_W4_TILE_PCT = 64 };
// -> the leading rows that go through the 8-wave kernel (a whole number of its rounds; 0: none, rows: all); the rest, if any, follow in
// a second launch of the 4-wave kernel over [rows8, rows).  Three candidates in units of one 256-row tile time / 100: every round in
// 8 waves; every round in 4 waves; the full 8-wave rounds + the remainder in 128-row tiles (+ X3_SPLIT_LAUNCH_PCT for the second launch):
// 98 304 rows (the fine pass of a 512-ray step) are 1.5 rounds of 256-row tiles = 2 tile times, or 1 + 0.64.
enum { X3_SPLIT_LAUNCH_PCT = 6 };
static inline int64_t x3_dgrad_rows8(int64_t rows) {
    const int64_t cus = num_cus(), round8 = 256 * cus, round4 = 128 * cus;
    const int64_t full8 = rows / round8, rem = rows - full8 * round8;
    const int64_t all8 = (rows + round8 - 1) / round8 * 100, all4 = (rows + round4 - 1) / round4 * X3_W4_TILE_PCT;
    const int64_t split = (full8 > 0 && rem > 0) ? full8 * 100 + (rem + round4 - 1) / round4 * X3_W4_TILE_PCT + X3_SPLIT_LAUNCH_PCT : INT64_MAX;
    if (split < all8 && split < all4) return full8 * round8;
    return all4 < all8 ? 0 : rows;
}
// the data-gradient launch(es) of a pass over the active rows [m.row_begin, m.rows): bf16x3 with plane areas picks its geometry per range
// (above); `pin` (measurement: sparf_launch_kernel 3 / 4) forces one geometry for the whole range
static int launch_dgrad(int prec, bool pose, bool q8, const MlpBwdArgs& m, hipStream_t s, int pin = 0) {
    const int64_t rows = m.rows - m.row_begin;
    if (prec != PREC_X3 || q8 || pin == 8) return launch_mlp_bwd(prec, pose, q8, m, mlp_grid(prec, rows), s, 8);
    const int64_t rows8 = pin == 4 ? 0 : x3_dgrad_rows8(rows);
    int rc = 0;
    if (rows8 > 0) {
        MlpBwdArgs a = m;
        a.rows = m.row_begin + rows8;
        rc = launch_mlp_bwd(prec, pose, q8, a, mlp_grid(prec, rows8), s, 8);
    }
    if (!rc && rows8 < rows) {
        MlpBwdArgs a = m;
        a.row_begin = m.row_begin + rows8;
        rc = launch_mlp_bwd(prec, pose, q8, a, mlp_grid(prec, rows - rows8), s, 4);
    }
    return rc;
}
static inline int wgrad_splits(int64_t rows, int* rows_per_split) {
    // ~4096 rows per split (measured: 2048 is slower for >= 256 k rows), but at least 25 splits when the
    // pass is small, so that the 10 jobs still fill the 256 CUs (65 k rows: 0.195 -> 0.148 ms) in ONE round of
    // workgroups (round 3 used 26: 260 workgroups, four of them a second round on their own -- the far rows'
    // 32 768-row fp32 passes of round 4 took 0.47 ms that way)
    // (round 4 re-measured the split size on the final kernels, same box: 3072 ... 49152 rows per split move the config-1 step by
    // <= 1 % (6.61-6.72 ms, inside the run-to-run spread) and 16384 costs configs 2 / 4 2-4 %: profiles/r04l_wgrad_rows_per_split.log)
    int64_t n = (rows + 4095) / 4096;
    const int64_t fill = rows / 512 < 25 ? rows / 512 : 25;
    if (n < fill) n = fill;
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    int64_t rps = ((rows + n - 1) / n + 63) / 64 * 64;
    if (rps < 64) rps = 64;                                   // rows == 0: one empty split
    n = (rows + rps - 1) / rps;
    if (n < 1) n = 1;
    *rows_per_split = (int)rps;
    return (int)n;
}

// the same split for at most `cap` splits (an active sub-range of a pass whose workspace holds `cap` partial blocks)
static inline void wgrad_splits_capped(int64_t rows, int cap, int* nsplit, int* rows_per_split) {
    int64_t n = cap < 1 ? 1 : cap;
    int64_t rps = ((rows + n - 1) / n + 63) / 64 * 64;
    if (rps < 64) rps = 64;
    n = (rows + rps - 1) / rps;
    if (n < 1) n = 1;
    *rows_per_split = (int)rps;
    *nsplit = (int)n;
}

// backward workspace layout
struct BwdWs {
    int64_t grad, d_sigma, d_z, d_len, partial, dp, dv, total;
    int nsplit, rows_per_split;
};
// far_count = K > 0: the last K samples of every ray go through far_prec as well (sparf_hip.h "far rows").  far_prec: fp32 only
// (row routing is compiled into the fp32 forward kernels, mlp_fwd_impl.h); main precision: a bf16-plane save layout (bf16, bf16x3)
// far_count = -1: far TILES by value (inference only): 128-row tiles whose largest depth sample exceeds far_thr go to far_prec, the
// others to prec; needs nsamp % 32 == 0 (a 32-row wave tile then lies inside one ray, whose samples increase)
static inline bool far_ok(int far_count, int far_prec, int nsamp, int prec) {
    if (far_count == 0) return true;
    if (far_prec != PREC_FP32 || prec == PREC_FP32 || nplanes_of(prec) != 1) return false;
    // tile routing decides per WORKGROUP tile (nwaves x 32 rows): both launches must cut the rows into the same tiles, or a tile of
    // the wider kernel that straddles the threshold would be skipped by it and only half-evaluated by the other (the
    // 8-wave bf16 kernels against the 4-wave fp32 far kernel; bf16x3 runs 4 waves)
    if (far_count == -1) return nsamp % 32 == 0 && nwaves_of(prec) == nwaves_of(far_prec);
    return far_count > 0 && far_count < nsamp;
}
static BwdWs bwd_ws_layout(int af, int nrays, int nsamp, int pose) {          // af: area format of the pass (layout.h)
    BwdWs w;
    const int64_t rows = (int64_t)nrays * nsamp;
    int64_t o = 0;
    w.grad = o; o += align256(grad_area_bytes(af, rows));
    w.d_sigma = o; o += align256(rows * 4);
    w.d_z = o; o += align256(rows * 12);
    w.d_len = o; o += align256((int64_t)nrays * 4);
    w.nsplit = wgrad_splits(rows, &w.rows_per_split);
    w.partial = o; o += align256((int64_t)w.nsplit * kPartialFloats * 4);
    w.dp = o; if (pose) o += align256(rows * 12);
    w.dv = o; if (pose) o += align256(rows * 128);
    w.total = o;
    return w;
}

// host segment array -> by-value kernel table; false if the segments do not tile [0, nrays) in order
static bool seg_table(int nseg, const sparf_segment_t* seg, int nrays, bool grads, SegTable* out) {
    out->n = 0;
    if (nseg == 0) return true;
    if (nseg < 0 || nseg > MAX_SEGMENTS || !seg) return false;
    int next = 0;
    for (int i = 0; i < nseg; ++i) {
        if (seg[i].ray0 != next || seg[i].nrays < 0) return false;
        next += seg[i].nrays;
        out->ray0[i] = seg[i].ray0;
        out->noise_scale[i] = seg[i].noise_scale;
        out->g_rgb[i] = grads ? seg[i].g_rgb : nullptr;
        out->g_depth[i] = grads ? seg[i].g_depth : nullptr;
        out->g_opacity[i] = grads ? seg[i].g_opacity : nullptr;
        out->g_weights[i] = grads ? seg[i].g_weights : nullptr;
        out->g_depth_var[i] = grads ? seg[i].g_depth_var : nullptr;
        out->g_rgb_var[i] = grads ? seg[i].g_rgb_var : nullptr;
        out->g_all_cum[i] = grads ? seg[i].g_all_cumulated : nullptr;
        out->g_density[i] = grads ? seg[i].g_density : nullptr;
        out->g_rgb_samples[i] = grads ? seg[i].g_rgb_samples : nullptr;
    }
    if (next != nrays) return false;
    // empty segments share their ray0 with the next one: the select chain keeps the LAST match, which is the non-empty one
    out->n = nseg;
    return true;
}

extern "C" {

int sparf_abi_version(void) { return SPARF_ABI_VERSION; }

int64_t sparf_table_count(int prec) { return prec_ok(prec) ? kTblCount[prec] : -1; }
int sparf_build_tables(int prec, int32_t* host_out) { return host_out ? build_tables(prec, host_out) : 1; }

int sparf_stream_nchunks(int prec, int backward) {
    if (!prec_ok(prec)) return -1;
    return backward ? bwd_nchunks(prec) : fwd_nchunks(prec);
}
int sparf_stream_chunk(int prec, int backward, int id, int32_t out[8]) {
    if (!prec_ok(prec) || !out || id < 0 || id >= sparf_stream_nchunks(prec, backward)) return 1;
    const Chunk c = backward ? bwd_chunk(prec, id) : fwd_chunk(prec, id);
    out[0] = c.layer; out[1] = c.seg; out[2] = c.mb0; out[3] = c.nmb; out[4] = c.ks0; out[5] = c.nks;
    out[6] = (int32_t)(backward ? bwd_chunk_off(prec, id) : fwd_chunk_off(prec, id));
    out[7] = chunk_bytes(prec, c);
    return 0;
}

int64_t sparf_packed_bytes(int prec) { return prec_ok(prec) ? kPackedBytes[prec] : -1; }
int sparf_pack_weights(int prec, const float* const* param_ptrs, const int32_t* tables, void* packed_out, void* stream) {
    if (!prec_ok(prec) || !param_ptrs || !tables || !packed_out) return 1;
    for (int i = 0; i < 2 * N_LAYERS; ++i)
        if (!param_ptrs[i]) return 1;
    return launch_pack(prec, param_ptrs, tables, packed_out, (hipStream_t)stream);
}

int sparf_c2f_weights(const float* progress, int has_c2f, float c2f_start, float c2f_end, float* out16, void* stream) {
    if (!out16 || (has_c2f && (!progress || !(c2f_end != c2f_start)))) return 1;
    return launch_c2f(progress, has_c2f, c2f_start, c2f_end, out16, (hipStream_t)stream);
}

int sparf_sample_coarse(const float* jitter, float u_const, const float* dmax_ray, const float* range_dev, float dmin, float scale,
                        int inverse, int nrays, int nsamp, float* t_out, void* stream) {
    if (nrays == 0 && nsamp > 0) return 0;
    if (nrays < 0 || nsamp <= 0 || !t_out) return 1;
    return launch_sample_coarse(jitter, u_const, dmax_ray, range_dev, dmin, scale, inverse, (int64_t)nrays * nsamp, nsamp, t_out,
                                (hipStream_t)stream);
}

int sparf_sample_fine(const float* weights, const float* t_coarse, const float* u_mid, const float* range_dev, float dmin, float dmax,
                      int nrays, int n_coarse, int n_fine, float* t_fine, float* t_out, void* stream) {
    if (nrays == 0 && n_coarse > 0 && n_fine > 0) return 0;
    if (nrays < 0 || n_coarse <= 0 || n_fine <= 0 || !weights || !t_coarse || !u_mid || !t_out) return 1;
    SampleFineArgs a{nrays, n_coarse, n_fine, weights, t_coarse, u_mid, dmin, dmax, range_dev, t_fine, t_out, {}};
    return launch_sample_fine(a, (hipStream_t)stream);
}

int sparf_sample_fine_hostgrid(const float* weights, const float* t_coarse, const float* u_mid_host, const float* range_dev, float dmin, float dmax,
                               int nrays, int n_coarse, int n_fine, float* t_fine, float* t_out, void* stream) {
    if (nrays == 0 && n_coarse > 0 && n_fine > 0) return 0;
    if (nrays < 0 || n_coarse <= 0 || n_fine <= 0 || n_fine > SAMPLE_FINE_HOST_MAX || !weights || !t_coarse || !u_mid_host || !t_out) return 1;
    SampleFineArgs a{nrays, n_coarse, n_fine, weights, t_coarse, nullptr, dmin, dmax, range_dev, t_fine, t_out, {}};
    for (int i = 0; i < n_fine; ++i) a.u_host[i] = u_mid_host[i];
    return launch_sample_fine(a, (hipStream_t)stream);
}

int sparf_ray_gen_forward(const float* pose, const float* intr, const float* pixels, const int64_t* ray_idx, int per_image,
                          int width, int nimg, int nrays, float* center, float* ray, void* stream) {
    if (nimg >= 0 && nrays == 0) return 0;                      // empty selection: nothing to write
    if (nimg < 0 || nrays < 0 || !pose || !intr || !center || !ray || ((pixels != nullptr) == (ray_idx != nullptr))) return 1;
    if (ray_idx && width <= 0) return 1;
    RayGenArgs a{nimg, nrays, width, per_image, pose, intr, pixels, ray_idx, center, ray};
    return launch_ray_gen_fwd(a, (hipStream_t)stream);
}

int sparf_ray_gen_backward(const float* pose, const float* intr, const float* pixels, const int64_t* ray_idx, int per_image,
                           int width, int nimg, int nrays, const float* d_center, const float* d_ray, float* d_pose,
                           void* stream) {
    if (nimg < 0 || nrays < 0 || !pose || !intr || !d_pose) return 1;
    if (nrays > 0 && ((pixels != nullptr) == (ray_idx != nullptr))) return 1;    // empty selection: d_pose = 0
    if (ray_idx && width <= 0) return 1;
    RayGenArgs a{nimg, nrays, width, per_image, pose, intr, pixels, ray_idx, nullptr, nullptr};
    return launch_ray_gen_bwd(a, d_center, d_ray, d_pose, (hipStream_t)stream);
}

int64_t sparf_adam_workspace_floats(void) { return 256; }

int sparf_adam_step(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace,
                    float* norm_out, float lr, float beta1, float beta2, float eps, int step, float max_norm, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || step < 1 || (max_norm > 0.0f && !workspace)) return 1;
    for (int i = 0; i < 2 * N_LAYERS; ++i)
        if (!params[i]) return 1;
    return launch_adam(params, grad, exp_avg, exp_avg_sq, workspace, norm_out, lr, beta1, beta2, eps, step, nullptr, max_norm, (hipStream_t)stream);
}

int sparf_adam_step_dev(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace,
                        float* norm_out, float lr, float beta1, float beta2, float eps, int* step_dev, float max_norm, void* stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev || (max_norm > 0.0f && !workspace)) return 1;
    for (int i = 0; i < 2 * N_LAYERS; ++i)
        if (!params[i]) return 1;
    return launch_adam(params, grad, exp_avg, exp_avg_sq, workspace, norm_out, lr, beta1, beta2, eps, 0, step_dev, max_norm, (hipStream_t)stream);
}

int sparf_photometric_loss(const float* pred, const float* pred_fine, const float* target, int64_t n, int kind, float delta,
                           float* loss, float* d_pred, float* d_pred_fine, float* workspace, void* stream) {
    if (n <= 0 || !pred || !target || !loss || (kind != 0 && kind != 1) || (kind == 1 && !(delta > 0.0f))) return 1;
    if (d_pred_fine && !pred_fine) return 1;
    return launch_photometric_loss(pred, pred_fine, target, n, kind, delta, loss, d_pred, d_pred_fine, workspace, (hipStream_t)stream);
}
int64_t sparf_photometric_workspace_floats(void) { return photometric_workspace_floats(); }

int64_t sparf_save_bytes(int prec, int64_t rows) {
    const PassPrec pp = pass_prec(prec);
    return pp.ok && rows >= 0 ? align256(save_area_bytes(pp.af, rows)) : -1;
}

int sparf_pass_forward(const sparf_pass_fwd_t* p, void* stream) {
    if (!p) return 1;
    const PassPrec pp = pass_prec(p->prec);
    const int prec = pp.base;
    if (!pp.ok || p->nrays < 0 || p->nsamp <= 0) return 1;
    if (pp.q8 && p->far_count > 0) return 1;           // far rows are transplanted into plane save areas only
    if (p->nrays == 0) return 0;
    const int64_t rows = (int64_t)p->nrays * p->nsamp;
    // one launch set takes up to 2^27 sample rows (the per-row outputs are indexed with 32-bit element offsets);
    // the save / gradient areas are addressed per 32-row tile block (layout.h) and have no limit of their own
    if (rows > ((int64_t)1 << 27)) return 4;
    if (!p->center || !p->dir || !p->t || !p->packed || !p->c2f || !p->venc_ws || !p->raylen || !p->sigma_raw || !p->rgb_samples ||
        !p->density || !p->weights || !p->rgb || !p->depth || !p->opacity || !p->depth_var || !p->rgb_var || !p->all_cumulated)
        return 1;
    if (!far_ok(p->far_count, p->far_prec, p->nsamp, prec)) return 1;
    if (p->far_count && (!p->far_packed || (p->save != nullptr && !p->far_ws) || (p->far_prec != prec && !p->far_venc_ws))) return 1;
    if (p->far_count == -1 && p->save != nullptr) return 1;            // tile routing: inference passes only
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_ray_setup(prec, p->dir, p->nrays, p->c2f + 10, p->venc_ws, p->raylen, s);
    if (rc) return rc;
    MlpFwdArgs m{(const char*)p->packed, p->c2f, p->center, p->dir, p->venc_ws, p->t, rows, p->nsamp, p->sigma_raw, p->rgb_samples, p->save};
    if (p->far_count == -1) { m.tile_thr = p->far_thr; m.tile_take = 1; }        // the tiles whose depth samples all stay below the threshold
    rc = launch_mlp_fwd(prec, !p->save ? FWD_INFER : pp.q8 ? FWD_SAVE_Q8 : FWD_SAVE_PLANES, m, mlp_grid(prec, rows), s);
    if (rc) return rc;
    if (p->far_count == -1) {
        // far tiles by value: the other tiles, through the far precision's inference kernel (every tile is evaluated by exactly one
        // of the two launches; the one that skips a tile spends four loads on it)
        rc = launch_ray_setup(p->far_prec, p->dir, p->nrays, p->c2f + 10, p->far_venc_ws, p->raylen, s);
        if (rc) return rc;
        MlpFwdArgs f{(const char*)p->far_packed, p->c2f, p->center, p->dir, p->far_venc_ws, p->t, rows, p->nsamp, p->sigma_raw, p->rgb_samples, nullptr};
        f.tile_thr = p->far_thr;
        f.tile_take = 2;
        rc = launch_mlp_fwd(p->far_prec, FWD_INFER, f, mlp_grid(p->far_prec, rows), s);
        if (rc) return rc;
    } else if (p->far_count) {
        // far rows: the last K samples of every ray once more, through the far precision's kernels, as a pass of nrays K-sample
        // rays whose outputs land on the main launch's (stream order: they replace what it wrote there); in a training pass what
        // the far launch saved -- activations and ReLU masks -- is then transplanted into the main save area, so that the backward
        // of the pass (bf16-operand arithmetic for every row) differentiates the forward that was actually composited
        const void* venc = p->venc_ws;
        if (p->far_prec != prec) {        // the per-ray view-encoding rows are laid out per precision (element type AND 16-byte-chunk order, ray_setup_kernel)
            rc = launch_ray_setup(p->far_prec, p->dir, p->nrays, p->c2f + 10, p->far_venc_ws, p->raylen, s);
            if (rc) return rc;
            venc = p->far_venc_ws;
        }
        const int64_t frows = (int64_t)p->nrays * p->far_count;
        MlpFwdArgs f{(const char*)p->far_packed, p->c2f, p->center, p->dir, venc, p->t, frows, p->far_count, p->sigma_raw, p->rgb_samples,
                     p->save ? p->far_ws : nullptr};
        f.row_stride = p->nsamp;
        f.row_off = p->nsamp - p->far_count;
        rc = launch_mlp_fwd(p->far_prec, p->save ? FWD_SAVE_PLANES : FWD_INFER, f, mlp_grid(p->far_prec, frows), s);
        if (rc) return rc;
        if (p->save) {
            rc = launch_far_transplant(prec, p->far_ws, p->save, frows, p->far_count, p->nsamp, s);
            if (rc) return rc;
        }
    }
    CompositeFwdArgs c{p->nrays, p->nsamp, p->t, p->sigma_raw, p->noise, p->noise_scale, p->rgb_samples, p->raylen, p->white_bg,
                       p->weights, p->density, p->rgb, p->depth, p->opacity, p->depth_var, p->rgb_var, p->all_cumulated, {}};
    if (!seg_table(p->nseg, p->seg, p->nrays, false, &c.seg)) return 1;
    return launch_composite_fwd(c, s);
}

int64_t sparf_bwd_workspace_bytes(int prec, int nrays, int nsamp, int pose) {
    const PassPrec pp = pass_prec(prec);
    if (!pp.ok || nrays < 0 || nsamp <= 0) return -1;
    return bwd_ws_layout(pp.af, nrays, nsamp, pose).total;
}
int sparf_pass_backward(const sparf_pass_bwd_t* p, void* stream) {
    if (!p) return 1;
    const PassPrec pp = pass_prec(p->prec);
    const int prec = pp.base;
    if (!pp.ok || p->nrays < 0 || p->nsamp <= 0) return 1;
    if (p->nrays == 0) {                                      // empty batch: zero parameter gradients
        if (!p->grad_params) return 1;
        return hipMemsetAsync(p->grad_params, 0, (size_t)N_PARAMS * sizeof(float), (hipStream_t)stream) == hipSuccess ? 0 : 2;
    }
    const int64_t rows = (int64_t)p->nrays * p->nsamp;
    if (rows > ((int64_t)1 << 27)) return 4;
    const bool pose = p->d_center != nullptr;
    if (pose != (p->d_dir != nullptr)) return 1;
    if (!p->center || !p->dir || !p->t || !p->packed || !p->c2f || !p->tables || !p->save || !p->raylen || !p->sigma_raw ||
        !p->rgb_samples || !p->weights || !p->ws || !p->grad_params)
        return 1;
    hipStream_t s = (hipStream_t)stream;
    const BwdWs w = bwd_ws_layout(pp.af, p->nrays, p->nsamp, pose);
    char* ws = (char*)p->ws;
    float* d_sigma = (float*)(ws + w.d_sigma);
    float* d_z = (float*)(ws + w.d_z);
    float* d_len = (float*)(ws + w.d_len);
    CompositeBwdArgs c{p->nrays, p->nsamp, p->t, p->sigma_raw, p->noise, p->noise_scale, p->rgb_samples, p->raylen, p->weights,
                       p->white_bg, p->g_rgb, p->g_depth, p->g_opacity, p->g_weights, d_sigma, d_z, pose ? d_len : nullptr, {}, 0};
    c.g_depth_var = p->g_depth_var; c.g_rgb_var = p->g_rgb_var; c.g_all_cum = p->g_all_cumulated;
    c.g_density = p->g_density; c.g_rgb_samples = p->g_rgb_samples;
    if (!seg_table(p->nseg, p->seg, p->nrays, true, &c.seg)) return 1;
    // Active ray range of a segmented pass: the segments that received an upstream gradient.  Rays are independent, so a
    // segment without one contributes nothing to any gradient; issued as separate calls autograd would not even call its
    // backward (the coarse pass of a correspondence render whose loss reads depth_fine only).  The backward kernels run
    // over the covering range [first active segment, last active segment] when that range starts on a 32-row tile.
    int ray0 = 0, ray1 = p->nrays;
    if (p->nseg > 0) {
        int first = -1, last = -1;
        for (int i = 0; i < p->nseg; ++i)
            if (p->seg[i].nrays > 0 && (p->seg[i].g_rgb || p->seg[i].g_depth || p->seg[i].g_opacity || p->seg[i].g_weights || p->seg[i].g_depth_var ||
                                        p->seg[i].g_rgb_var || p->seg[i].g_all_cumulated || p->seg[i].g_density || p->seg[i].g_rgb_samples)) {
                if (first < 0) first = i;
                last = i;
            }
        if (first < 0) {                                          // no gradient at all: zero results
            if (hipMemsetAsync(p->grad_params, 0, (size_t)N_PARAMS * sizeof(float), s) != hipSuccess) return 2;
            if (pose && !p->accumulate_rays && (hipMemsetAsync(p->d_center, 0, (size_t)p->nrays * 12, s) != hipSuccess ||
                         hipMemsetAsync(p->d_dir, 0, (size_t)p->nrays * 12, s) != hipSuccess)) return 2;
            return 0;
        }
        ray0 = p->seg[first].ray0;
        ray1 = p->seg[last].ray0 + p->seg[last].nrays;
        if (((int64_t)ray0 * p->nsamp) % 32 != 0) ray0 = 0;       // (64 / 192 samples per ray: always aligned)
    }
    const int64_t row0 = (int64_t)ray0 * p->nsamp, row1 = (int64_t)ray1 * p->nsamp;
    c.ray_base = ray0;
    c.nrays = ray1 - ray0;
    int rc = launch_composite_bwd(c, s);
    if (rc) return rc;
    MlpBwdArgs m{(const char*)p->packed, p->c2f, p->center, p->dir, p->t, row1, p->nsamp, p->save, ws + w.grad, d_sigma, d_z,
                 (float*)(ws + w.dp), (float*)(ws + w.dv), row0, rows};
    // split count of the active range.  wgrad_splits is NOT monotone in its row count (12289 rays x 64 samples: 127 splits,
    // a sub-range of 8684 rays: 128), and `partial` was sized for the whole pass: never more splits than the workspace holds
    int rps = w.rows_per_split;
    int nsplit = w.nsplit;
    if (!(row0 == 0 && row1 == rows)) {
        nsplit = wgrad_splits(row1 - row0, &rps);
        if (nsplit > w.nsplit) wgrad_splits_capped(row1 - row0, w.nsplit, &nsplit, &rps);
    }
    // (A chunked schedule -- dgrad of row range c on this stream with a reduced grid, wgrad of range c-1 on a side stream on the CUs
    // left, matrix-pipe-bound against HBM-bound -- was built and measured in round 4: bit-identical gradients, 3.5-13 % SLOWER than
    // this serial order at 2-8 chunks and 32-96 reserved CUs, profiles/r04e_overlap_schedule_sweep.log.  Removed.)
    rc = launch_dgrad(prec, pose, pp.q8, m, s);
    if (rc) return rc;
    WgradArgs g{p->save, ws + w.grad, row1, rps, (float*)(ws + w.partial), row0};
    rc = launch_wgrad(prec, pp.q8, g, nsplit, p->tables + kWsrcOff[prec], p->grad_params, s);
    if (rc) return rc;
    if (pose) {
        const float* c2f_view = p->c2f + 10;
        // rays outside the active range receive no gradient: zero them unless the caller accumulates onto an earlier pass's
        if (!p->accumulate_rays) {
            if (ray0 > 0 && (hipMemsetAsync(p->d_center, 0, (size_t)ray0 * 12, s) != hipSuccess || hipMemsetAsync(p->d_dir, 0, (size_t)ray0 * 12, s) != hipSuccess)) return 2;
            if (ray1 < p->nrays && (hipMemsetAsync(p->d_center + (size_t)ray1 * 3, 0, (size_t)(p->nrays - ray1) * 12, s) != hipSuccess ||
                                    hipMemsetAsync(p->d_dir + (size_t)ray1 * 3, 0, (size_t)(p->nrays - ray1) * 12, s) != hipSuccess)) return 2;
        }
        RayReduceArgs r{ray1 - ray0, p->nsamp, p->t, (const float*)(ws + w.dp), (const float*)(ws + w.dv), p->dir, p->raylen, d_len,
                        c2f_view, p->d_center, p->d_dir, ray0};
        r.accumulate = p->accumulate_rays;
        rc = launch_ray_reduce(r, s);
    }
    return rc;
}

// ---- stand-alone compositing (sparf_hip.h): the compositing kernels of a pass on caller-built per-sample values
int sparf_composite_forward(const sparf_composite_fwd_t* p, void* stream) {
    if (!p || p->nrays < 0 || p->nsamp <= 0) return 1;
    if (p->nrays == 0) return 0;
    if ((int64_t)p->nrays * p->nsamp > ((int64_t)1 << 27)) return 4;
    if (!p->dir || !p->t || !p->density || !p->rgb_samples || !p->raylen || !p->weights || !p->rgb || !p->depth || !p->opacity || !p->depth_var ||
        !p->rgb_var || !p->all_cumulated)
        return 1;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_ray_setup(-1, p->dir, p->nrays, nullptr, nullptr, p->raylen, s);           // |dir| only
    if (rc) return rc;
    CompositeFwdArgs c{p->nrays, p->nsamp, p->t, p->density, nullptr, 0.0f, p->rgb_samples, p->raylen, p->white_bg,
                       p->weights, nullptr, p->rgb, p->depth, p->opacity, p->depth_var, p->rgb_var, p->all_cumulated, {}};
    c.seg.n = 0;
    c.direct = 1;
    return launch_composite_fwd(c, s);
}
int sparf_composite_backward(const sparf_composite_bwd_t* p, void* stream) {
    if (!p || p->nrays < 0 || p->nsamp <= 0) return 1;
    if (p->nrays == 0) return 0;
    if ((int64_t)p->nrays * p->nsamp > ((int64_t)1 << 27)) return 4;
    if (!p->dir || !p->t || !p->density || !p->rgb_samples || !p->raylen || !p->weights || !p->d_density || !p->d_rgb_samples) return 1;
    if (p->d_dir && !p->d_len_ws) return 1;
    hipStream_t s = (hipStream_t)stream;
    CompositeBwdArgs c{p->nrays, p->nsamp, p->t, p->density, nullptr, 0.0f, p->rgb_samples, p->raylen, p->weights,
                       p->white_bg, p->g_rgb, p->g_depth, p->g_opacity, p->g_weights, p->d_density, p->d_rgb_samples, p->d_dir ? p->d_len_ws : nullptr, {}, 0};
    c.seg.n = 0;
    c.g_depth_var = p->g_depth_var; c.g_rgb_var = p->g_rgb_var; c.g_all_cum = p->g_all_cumulated;
    c.direct = 1;
    int rc = launch_composite_bwd(c, s);
    if (rc) return rc;
    if (p->d_dir) rc = launch_len_to_dir(p->dir, p->raylen, p->d_len_ws, p->nrays, p->d_dir, s);
    return rc;
}

// the wgrad split of a pass of `rows_total` rows restricted to an active range of `rows_active` rows (host arithmetic only;
// tests/test_tables_cpu.py checks that it never exceeds what sparf_bwd_workspace_bytes reserved)
int sparf_debug_wgrad_split(int64_t rows_total, int64_t rows_active, int* nsplit_total, int* nsplit_active, int* rows_per_split_active) {
    if (rows_total < 0 || rows_active < 0 || rows_active > rows_total || !nsplit_total || !nsplit_active || !rows_per_split_active) return 1;
    int rps = 0;
    const int cap = wgrad_splits(rows_total, &rps);
    int n = cap;
    if (rows_active != rows_total) {
        n = wgrad_splits(rows_active, &rps);
        if (n > cap) wgrad_splits_capped(rows_active, cap, &n, &rps);
    }
    *nsplit_total = cap; *nsplit_active = n; *rows_per_split_active = rps;
    return 0;
}

int sparf_launch_kernel(int which, const sparf_pass_fwd_t* f, const sparf_pass_bwd_t* b, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) {
        if (!f || !pass_prec(f->prec).ok || !f->venc_ws || !f->c2f) return 1;
        const PassPrec fp = pass_prec(f->prec);
        const int64_t rows = (int64_t)f->nrays * f->nsamp;
        MlpFwdArgs m{(const char*)f->packed, f->c2f, f->center, f->dir, f->venc_ws, f->t, rows, f->nsamp, f->sigma_raw, f->rgb_samples, f->save};
        return launch_mlp_fwd(fp.base, !f->save ? FWD_INFER : fp.q8 ? FWD_SAVE_Q8 : FWD_SAVE_PLANES, m, mlp_grid(fp.base, rows), s);
    }
    if (!b || !pass_prec(b->prec).ok || !b->ws) return 1;
    const PassPrec bp = pass_prec(b->prec);
    const int64_t rows = (int64_t)b->nrays * b->nsamp;
    const bool pose = b->d_center != nullptr;
    const BwdWs w = bwd_ws_layout(bp.af, b->nrays, b->nsamp, pose);
    char* ws = (char*)b->ws;
    if (which == 1 || which == 3 || which == 4) {        // 3 / 4: the bf16x3 data-gradient kernel pinned to its 8-wave / 4-wave geometry (measurement)
        MlpBwdArgs m{(const char*)b->packed, b->c2f, b->center, b->dir, b->t, rows, b->nsamp, b->save, ws + w.grad, (float*)(ws + w.d_sigma),
                     (float*)(ws + w.d_z), (float*)(ws + w.dp), (float*)(ws + w.dv), 0, rows};
        return launch_dgrad(bp.base, pose, bp.q8, m, s, which == 1 ? 0 : which == 3 ? 8 : 4);
    }
    if (which == 2) {
        WgradArgs g{b->save, ws + w.grad, rows, w.rows_per_split, (float*)(ws + w.partial)};
        return launch_wgrad(bp.base, bp.q8, g, w.nsplit, b->tables + kWsrcOff[bp.base], b->grad_params, s);
    }
    return 1;
}

// ---- calibration (measurement only, sparf_hip.h): fixed kernels that do not change with the renderer's
int64_t sparf_calib_mfma(int iters, float* sink, void* stream) {
    if (iters <= 0 || iters > (1 << 24) || !sink) return -1;
    const int grid = num_cus();
    if ((int64_t)grid * 512 > SPARF_CALIB_SINK_FLOATS) return -1;
    if (launch_calib_mfma(iters, sink, grid, (hipStream_t)stream)) return -2;
    return calib_mfma_flops(iters, grid);
}
int sparf_calib_hbm(const void* src, void* dst, int64_t bytes, int mode, float* sink, void* stream) {
    if (!src || bytes < 1024 || (bytes & 1023) || (mode != 0 && mode != 1) || (mode == 1 && !dst) || !sink) return 1;
    return launch_calib_hbm(src, dst, bytes, mode, sink, 4 * num_cus(), (hipStream_t)stream);
}

}  // extern "C"
