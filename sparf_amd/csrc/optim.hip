// Optimiser step for one NeRF network on its flat gradient buffer:
// gradient-norm clipping (torch.nn.utils.clip_grad_norm_, /root/reference/source/training/
// base.py:96-97 via engine after_backward) + Adam (torch.optim.Adam as configured by
// /root/reference/source/training/nerf_trainer.py:181-185: betas (0.9, 0.999), eps 1e-8, no
// weight decay, no amsgrad) in two launches instead of ~15 multi-tensor kernels.
//
// The backward kernels deliver all 20 parameter gradients of a network as ONE flat fp32
// buffer in (W0, b0, ..., W9, b9) order (layout.h param_w_off / param_b_off); the
// parameters themselves stay ordinary nn.Linear tensors, addressed through a pointer table.
#include "kernels.h"
#include "layout.h"

namespace sparf {

enum { OPT_BLOCK = 256, OPT_PARTS = 256 };

// stage 1 of ||g||^2: OPT_PARTS fixed-order partial sums (deterministic).  With a device-side update counter
// (sparf_adam_step_dev: a captured hipGraph replays the SAME launch arguments every step, so the step number
// behind Adam's bias corrections cannot be a host integer) this launch also advances the counter; the Adam
// launch that follows on the stream reads it.
__global__ void __launch_bounds__(OPT_BLOCK) grad_sqnorm_kernel(const float* __restrict__ g, int n, float* __restrict__ parts, int* __restrict__ step_dev) {
    if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) *step_dev += 1;
    if (!parts) return;
    float s = 0.f;
    for (int i = blockIdx.x * OPT_BLOCK + threadIdx.x; i < n; i += OPT_PARTS * OPT_BLOCK) s += g[i] * g[i];
    __shared__ float red[OPT_BLOCK];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = OPT_BLOCK / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}

struct OptPtrs { float* p[2 * N_LAYERS]; };      // W0, b0, W1, b1, ...

struct AdamArgs {
    OptPtrs p;                 // 20 parameter tensors (updated in place)
    const float* grad;         // [N_PARAMS] flat gradient
    float* exp_avg;            // [N_PARAMS]
    float* exp_avg_sq;         // [N_PARAMS]
    const float* parts;        // OPT_PARTS partial squared norms, or nullptr (no clipping)
    float* norm_out;           // total gradient norm before clipping (optional)
    float lr, beta1, beta2, eps, bias1, bias2_sqrt, max_norm;
    const int* step_dev;       // device-side update count (already advanced for this step) or nullptr: bias1 / bias2_sqrt from the host
};

__global__ void __launch_bounds__(OPT_BLOCK) adam_kernel(AdamArgs a) {
    // pointer table and offsets in LDS: a by-value kernel-argument array indexed with a
    // runtime value would be demoted to scratch memory
    __shared__ int off[2 * N_LAYERS + 1];
    __shared__ float* ptr[2 * N_LAYERS];
    __shared__ float clip, sh_bias1, sh_bias2_sqrt;
    if (threadIdx.x == 1) {
        float b1 = a.bias1, b2s = a.bias2_sqrt;
        if (a.step_dev) {
            const double st = (double)*a.step_dev;
            b1 = (float)(1.0 - pow((double)a.beta1, st));
            b2s = (float)sqrt(1.0 - pow((double)a.beta2, st));
        }
        sh_bias1 = b1; sh_bias2_sqrt = b2s;
    }
    if (threadIdx.x < 2 * N_LAYERS + 1) {
        const int k = threadIdx.x;
        off[k] = k == 2 * N_LAYERS ? (int)N_PARAMS : (int)((k & 1) ? param_b_off(k >> 1) : param_w_off(k >> 1));
    }
#pragma unroll
    for (int k = 0; k < 2 * N_LAYERS; ++k)
        if ((int)threadIdx.x == k + 32) ptr[k] = a.p.p[k];
    if (threadIdx.x == 0) {
        float c = 1.0f;
        if (a.parts) {
            float s = 0.f;
            for (int i = 0; i < OPT_PARTS; ++i) s += a.parts[i];
            const float norm = sqrtf(s);
            c = fminf(a.max_norm / (norm + 1e-6f), 1.0f);        // torch: clip_coef clamped to 1
            if (a.norm_out && blockIdx.x == 0) *a.norm_out = norm;
        }
        clip = c;
    }
    __syncthreads();
    const int i = blockIdx.x * OPT_BLOCK + threadIdx.x;
    if (i >= N_PARAMS) return;
    int k = 0;
#pragma unroll
    for (int step = 16; step > 0; step >>= 1)
        if (k + step <= 2 * N_LAYERS && off[k + step] <= i) k += step;
    float* w = ptr[k] + (i - off[k]);
    const float g = a.grad[i] * clip;
    const float m = a.exp_avg[i] + (1.0f - a.beta1) * (g - a.exp_avg[i]);           // lerp, as torch
    const float v = a.beta2 * a.exp_avg_sq[i] + (1.0f - a.beta2) * g * g;
    a.exp_avg[i] = m;
    a.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / sh_bias2_sqrt + a.eps;
    *w = *w - (a.lr / sh_bias1) * (m / denom);
}

int launch_adam(const float* const* params, const float* grad, float* exp_avg, float* exp_avg_sq, float* workspace, float* norm_out,
                float lr, float beta1, float beta2, float eps, int step, int* step_dev, float max_norm, hipStream_t s) {
    AdamArgs a;
    for (int i = 0; i < 2 * N_LAYERS; ++i) a.p.p[i] = const_cast<float*>(params[i]);
    a.grad = grad; a.exp_avg = exp_avg; a.exp_avg_sq = exp_avg_sq;
    a.parts = nullptr; a.norm_out = norm_out;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_norm;
    a.step_dev = step_dev;
    a.bias1 = (float)(1.0 - pow((double)beta1, (double)(step > 0 ? step : 1)));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)(step > 0 ? step : 1)));
    if (max_norm > 0.0f || step_dev) {
        hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(max_norm > 0.0f ? OPT_PARTS : 1), dim3(OPT_BLOCK), 0, s, grad, (int)N_PARAMS,
                           max_norm > 0.0f ? workspace : (float*)nullptr, step_dev);
        if (max_norm > 0.0f) a.parts = workspace;
    }
    hipLaunchKernelGGL(adam_kernel, dim3((N_PARAMS + OPT_BLOCK - 1) / OPT_BLOCK), dim3(OPT_BLOCK), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}


// ---- photometric loss + gradient seed ------------------------------------------------------
// BaseLoss.MSE_loss / huber_loss of /root/reference/source/training/core/base_losses.py:151-156
// applied to rgb and (optionally) rgb_fine against the same target, summed
// (base_losses.py:303-311):   kind 0: sum((p-t)^2) / (n + 1e-6)      kind 1: 2 * mean(huber_delta(p - t))
// Fixed-order reductions (deterministic): up to PHOTO_PARTS workgroups each write the loss gradient of
// their grid-stride share and one partial sum, the last launch adds the partials in index order.
// Small inputs (a 4096-ray batch is 12 k elements) take the single-workgroup path in one launch.
enum { PHOTO_BLOCK = 1024, PHOTO_PARTS = 256, PHOTO_SINGLE_MAX = 65536 };

__global__ void __launch_bounds__(PHOTO_BLOCK) photometric_loss_kernel(const float* __restrict__ pred, const float* __restrict__ pred_fine,
                                                                       const float* __restrict__ target, int64_t n, int kind, float delta,
                                                                       float* __restrict__ out, float* __restrict__ d_pred,
                                                                       float* __restrict__ d_pred_fine, int final_scale) {
    const float inv = kind == 0 ? (float)(1.0 / ((double)n + 1e-6)) : (float)(2.0 / (double)n);
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * PHOTO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PHOTO_BLOCK) {
        const float t = target[i];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float* p = f ? pred_fine : pred;
            float* d = f ? d_pred_fine : d_pred;
            if (!p) continue;
            const float e = p[i] - t;
            float l, g;
            if (kind == 0) { l = e * e; g = 2.0f * e; }
            else if (fabsf(e) <= delta) { l = 0.5f * e * e; g = e; }
            else { l = delta * (fabsf(e) - 0.5f * delta); g = e > 0.f ? delta : -delta; }
            s += l;
            if (d) d[i] = g * inv;
        }
    }
    __shared__ float red[PHOTO_BLOCK];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = PHOTO_BLOCK / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = final_scale ? red[0] * inv : red[0];
}

__global__ void __launch_bounds__(PHOTO_PARTS) photometric_final_kernel(const float* __restrict__ parts, int nparts, int64_t n, int kind,
                                                                        float* __restrict__ loss) {
    __shared__ float red[PHOTO_PARTS];
    red[threadIdx.x] = (int)threadIdx.x < nparts ? parts[threadIdx.x] : 0.f;
    __syncthreads();
    for (int k = PHOTO_PARTS / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * (kind == 0 ? (float)(1.0 / ((double)n + 1e-6)) : (float)(2.0 / (double)n));
}

int photometric_workspace_floats() { return PHOTO_PARTS; }

int launch_photometric_loss(const float* pred, const float* pred_fine, const float* target, int64_t n, int kind, float delta,
                            float* loss, float* d_pred, float* d_pred_fine, float* workspace, hipStream_t s) {
    if (n <= PHOTO_SINGLE_MAX || !workspace) {
        hipLaunchKernelGGL(photometric_loss_kernel, dim3(1), dim3(PHOTO_BLOCK), 0, s, pred, pred_fine, target, n, kind, delta, loss,
                           d_pred, d_pred_fine, 1);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    int64_t nb = (n + 4 * PHOTO_BLOCK - 1) / (4 * PHOTO_BLOCK);
    if (nb > PHOTO_PARTS) nb = PHOTO_PARTS;
    hipLaunchKernelGGL(photometric_loss_kernel, dim3((int)nb), dim3(PHOTO_BLOCK), 0, s, pred, pred_fine, target, n, kind, delta, workspace,
                       d_pred, d_pred_fine, 0);
    if (hipGetLastError() != hipSuccess) return 2;
    hipLaunchKernelGGL(photometric_final_kernel, dim3(1), dim3(PHOTO_PARTS), 0, s, workspace, (int)nb, n, kind, loss);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
