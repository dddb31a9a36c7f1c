// Per-ray kernels around the fused MLP: view-direction encoding, stratified depth samples,
// alpha compositing (forward / backward), inverse-CDF resampling + merge sort, and the
// reduction of per-sample point gradients to ray origin / direction gradients.
//
// Reference semantics:
//   /root/reference/source/models/renderer.py:383-456, 595-624   (sampling)
//   /root/reference/source/models/frequency_nerf.py:199-211, 283-343 (view enc, composite)
// One wavefront (64 lanes) owns one ray; prefix sums along the ray are wave scans carried
// in fp64 (torch's CPU cumsum accumulates float data in double, and the oracle is torch
// CPU; fp64 also removes scan-order sensitivity).
#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

static SP_DEV double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
static SP_DEV double wave_rev_incl_scan(double v, int lane) {      // sum over lanes >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_down(v, d);
        if (lane + d < 64) v += o;
    }
    return v;
}
static SP_DEV double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
static SP_DEV float wave_sumf(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
static SP_DEV float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // torch F.softplus(beta=1, threshold=20)
static SP_DEV float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------ ray setup
// venc[ray][32] (stage_t, pos layout): [d(3), 4 sin, 4 cos per coord] of d = ray/max(|ray|,1e-12)
// with the BARF band mask; raylen[ray] = |ray|.
template <int PREC>
__global__ void ray_setup_kernel(const float* __restrict__ dir, int nrays, const float* __restrict__ c2f_view,
                                 typename Policy<PREC>::stage_t* __restrict__ venc, float* __restrict__ raylen) {
    typedef typename Policy<PREC>::stage_t act_t;
    constexpr int CH = Policy<PREC>::CH;
    int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= nrays) return;
    float x = dir[ray * 3], y = dir[ray * 3 + 1], z = dir[ray * 3 + 2];
    float len = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
    raylen[ray] = len;
    float inv = fmaxf(len, 1e-12f);
    float d[3] = {x / inv, y / inv, z / inv};
    float feat[V_DIM];
    feat[0] = d[0]; feat[1] = d[1]; feat[2] = d[2];
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < LVIEW; ++k) {
            float s, co;
            sincosf(__fmul_rn(d[c], pe_freq(k)), &s, &co);
            feat[3 + c * 8 + k] = __fmul_rn(s, c2f_view[k]);
            feat[3 + c * 8 + 4 + k] = __fmul_rn(co, c2f_view[k]);
        }
    act_t* o = venc + (int64_t)ray * 32;
    for (int h = 0; h < 2; ++h)
        for (int q = 0; q < 16; ++q) {
            int f = view_feat(q, h);
            o[pos_of(q, h, CH)] = (act_t)(f < 0 ? 0.0f : feat[f]);
        }
}

// ------------------------------------------------------------------ coarse depth samples
// t = (u + i)/N * scale + dmin ; optionally t = 1/(t + 1e-8)   (renderer.py:404-416)
// jitter == nullptr -> u = u_const (0.5 for deterministic modes, 1.0 for render_to_max)
// dmax_ray != nullptr -> scale = dmax_ray[ray] - dmin           (renderer.py:616-621)
// range_dev != nullptr -> dmin = range_dev[0] and (without dmax_ray) scale = range_dev[1] - range_dev[0], an fp32
// subtraction as torch does for tensor ranges (trainers pass data_dict.depth_range[0], a device tensor): no host readback
__global__ void sample_coarse_kernel(const float* __restrict__ jitter, float u_const, const float* __restrict__ dmax_ray,
                                     const float* __restrict__ range_dev, float dmin, float scale, int inverse, int64_t rows,
                                     int nsamp, float* __restrict__ t) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    if (range_dev) {
        dmin = range_dev[0];
        if (!dmax_ray) scale = __fsub_rn(range_dev[1], dmin);
    }
    int s = (int)(i % nsamp);
    float u = jitter ? jitter[i] : u_const;
    float sc = dmax_ray ? __fsub_rn(dmax_ray[i / nsamp], dmin) : scale;
    float v = __fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(u, (float)s), (float)nsamp), sc), dmin);
    if (inverse) v = __fdiv_rn(1.0f, __fadd_rn(v, 1e-8f));
    t[i] = v;
}

// ------------------------------------------------------------------ ray segments
// the segment that holds `ray` (wave-uniform): last entry with ray0 <= ray, found by an unrolled select chain
struct SegPick {
    int ray0; float noise_scale; const float *g_rgb, *g_depth, *g_opacity, *g_weights;
    const float *g_depth_var, *g_rgb_var, *g_all_cum, *g_density, *g_rgb_samples;
};
static SP_DEV SegPick pick_segment(const SegTable& st, int ray, const SegPick& pass_level) {
    SegPick p = pass_level;
    if (st.n > 0) {
#pragma unroll
        for (int s = 0; s < MAX_SEGMENTS; ++s)
            if (s < st.n && ray >= st.ray0[s])
                p = SegPick{st.ray0[s], st.noise_scale[s], st.g_rgb[s], st.g_depth[s], st.g_opacity[s], st.g_weights[s],
                            st.g_depth_var[s], st.g_rgb_var[s], st.g_all_cum[s], st.g_density[s], st.g_rgb_samples[s]};
    }
    return p;
}

// ------------------------------------------------------------------ compositing, forward
__global__ void __launch_bounds__(64) composite_fwd_kernel(CompositeFwdArgs a) {
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int N = a.nsamp;
    const int64_t base = (int64_t)ray * N;
    const float ell = a.raylen[ray];
    const float noise_scale = pick_segment(a.seg, ray, SegPick{0, a.noise_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}).noise_scale;
    double carry = 0.0;            // sum of sigma*delta over all previous samples
    float s_w = 0.f, s_d = 0.f, s_r = 0.f, s_g = 0.f, s_b = 0.f;
    float T_nm2 = 1.0f;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int i = j0 + lane;
        const bool ok = i < N;
        float sd = 0.f, tt = 0.f, dens = 0.f;
        if (ok) {
            tt = a.t[base + i];
            float tn = i + 1 < N ? a.t[base + i + 1] : 0.f;
            float delta = i + 1 < N ? __fsub_rn(tn, tt) : 1e10f;
            float raw = a.sigma_raw[base + i];
            if (a.noise && noise_scale != 0.0f) raw = __fadd_rn(raw, __fmul_rn(a.noise[base + i], noise_scale));
            dens = a.direct ? raw : softplus_f(raw);
            sd = __fmul_rn(dens, __fmul_rn(delta, ell));
        }
        double incl = wave_incl_scan((double)sd, lane);
        double excl = carry + incl - (double)sd;
        carry += __shfl(incl, 63);
        if (ok) {
            float T = expf(-(float)excl);
            float alpha = 1.0f - expf(-sd);
            float w = T * alpha;
            a.weights[base + i] = w;
            if (a.density) a.density[base + i] = dens;
            if (i == N - 2) T_nm2 = T;
            const float* c = a.rgb_samples + (base + i) * 3;
            s_w += w; s_d += w * tt; s_r += w * c[0]; s_g += w * c[1]; s_b += w * c[2];
        }
    }
    const float opacity = wave_sumf(s_w), depth = wave_sumf(s_d);
    const float r = wave_sumf(s_r), g = wave_sumf(s_g), b = wave_sumf(s_b);
    // T at index N-2 lives in one lane: broadcast by max (T <= 1, others hold... use sum trick)
    float Tn = (N >= 2 && ((N - 2) % 64) == lane) ? T_nm2 : 0.f;
    Tn = wave_sumf(Tn);
    float v_d = 0.f, v_c = 0.f;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int i = j0 + lane;
        if (i < N) {
            float w = a.weights[base + i], tt = a.t[base + i];
            const float* c = a.rgb_samples + (base + i) * 3;
            float dd = tt - depth;
            v_d += w * dd * dd;
            v_c += w * ((c[0] - r) + (c[1] - g) + (c[2] - b));
        }
    }
    v_d = wave_sumf(v_d); v_c = wave_sumf(v_c);
    if (lane == 0) {
        float add = a.white_bg ? 1.0f - opacity : 0.0f;
        a.rgb[ray * 3 + 0] = r + add; a.rgb[ray * 3 + 1] = g + add; a.rgb[ray * 3 + 2] = b + add;
        a.depth[ray] = depth; a.opacity[ray] = opacity; a.depth_var[ray] = v_d; a.rgb_var[ray] = v_c;
        a.all_cumulated[ray] = Tn;
    }
}

// ------------------------------------------------------------------ compositing, backward
// q_i = dL/dw_i = gC.c_i + gD t_i + gO + gW_i  [+ the depth_var / rgb_var terms] ; dL/ds_j = T_{j+1} q_j - sum_{i>j} w_i q_i
// The other outputs of NeRF.composite (frequency_nerf.py:317-338; plain autograd in the reference, ABI 6 here), with D = depth,
// O = opacity, S_i = sum_ch c_i, SC = sum_i w_i S_i (= the rendered colour summed over channels, before the background):
//   depth_var  V = sum w (t - D)^2       dV/dw_i = (t_i - D)^2 - 2 t_i D (1 - O)            (D moves with w_i: sum_j w_j (t_j - D) = D - D O)
//   rgb_var    U = sum_i w_i (S_i - SC) = SC (1 - O)     dU/dw_i = S_i (1 - O) - SC,   dU/dc_i,ch = w_i (1 - O)
//   all_cumulated A = T_{N-2} = exp(-sum_{j<N-2} s_j)    dA/ds_j = -A for j <= N-3
//   density, rgb_samples: added onto d density_i, d c_i directly.
__global__ void __launch_bounds__(64) composite_bwd_kernel(CompositeBwdArgs a) {
    const int ray = a.ray_base + blockIdx.x, lane = threadIdx.x;
    const int N = a.nsamp;
    const int64_t base = (int64_t)ray * N;
    const float ell = a.raylen[ray];
    const SegPick sg = pick_segment(a.seg, ray, SegPick{0, a.noise_scale, a.g_rgb, a.g_depth, a.g_opacity, a.g_weights,
                                                      a.g_depth_var, a.g_rgb_var, a.g_all_cum, a.g_density, a.g_rgb_samples});
    const int lray = ray - sg.ray0;                   // ray index inside its segment's gradient tensors
    const float noise_scale = sg.noise_scale;
    const float* gw = sg.g_weights ? sg.g_weights + (int64_t)lray * N : nullptr;
    const float* gdn = sg.g_density ? sg.g_density + (int64_t)lray * N : nullptr;
    const float* grs = sg.g_rgb_samples ? sg.g_rgb_samples + (int64_t)lray * N * 3 : nullptr;
    float gC[3] = {0.f, 0.f, 0.f}, gD = 0.f, gO = 0.f;
    if (sg.g_rgb) { gC[0] = sg.g_rgb[lray * 3]; gC[1] = sg.g_rgb[lray * 3 + 1]; gC[2] = sg.g_rgb[lray * 3 + 2]; }
    if (sg.g_depth) gD = sg.g_depth[lray];
    if (sg.g_opacity) gO = sg.g_opacity[lray];
    if (a.white_bg) gO -= gC[0] + gC[1] + gC[2];
    const float gV = sg.g_depth_var ? sg.g_depth_var[lray] : 0.f, gU = sg.g_rgb_var ? sg.g_rgb_var[lray] : 0.f;
    const float gA = sg.g_all_cum ? sg.g_all_cum[lray] : 0.f;
    // the ray's own depth / opacity / colour sum, needed by the variance terms only (wave-uniform branch)
    float Dr = 0.f, Or = 0.f, SC = 0.f;
    if (sg.g_depth_var || sg.g_rgb_var) {
        for (int i = lane; i < N; i += 64) {
            const float w = a.weights[base + i];
            const float* c = a.rgb_samples + (base + i) * 3;
            Dr += w * a.t[base + i]; Or += w; SC += w * (c[0] + c[1] + c[2]);
        }
        Dr = wave_sumf(Dr); Or = wave_sumf(Or); SC = wave_sumf(SC);
    }
    // pass 1 (forward along the ray): T_{i+1} = exp(-sum_{k<=i} s_k), parked in d_sigma_raw[i] (the same lane
    // reads it back in pass 2)
    double carry_sd = 0.0;
    float A_part = 0.f;            // T_{N-2} = the parked value of sample N-3 (1 for N == 2)
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int i = j0 + lane;
        float sd = 0.f;
        if (i < N) {
            const float tt = a.t[base + i];
            const float delta = i + 1 < N ? __fsub_rn(a.t[base + i + 1], tt) : 1e10f;
            float raw = a.sigma_raw[base + i];
            if (a.noise && noise_scale != 0.0f) raw = __fadd_rn(raw, __fmul_rn(a.noise[base + i], noise_scale));
            sd = __fmul_rn(a.direct ? raw : softplus_f(raw), __fmul_rn(delta, ell));
        }
        double incl_sd = carry_sd + wave_incl_scan((double)sd, lane);
        carry_sd = __shfl(incl_sd, 63);
        if (i < N) {
            const float Tn = expf(-(float)incl_sd);
            a.d_sigma_raw[base + i] = Tn;
            if (i == N - 3) A_part = Tn;
        }
    }
    const float A = sg.g_all_cum ? (N >= 3 ? wave_sumf(A_part) : N == 2 ? 1.0f : 0.0f) : 0.0f;
    // pass 2 (backward along the ray): suffix_i = sum_{k>i} w_k q_k accumulated FROM THE FAR END, so that it is
    // exactly 0 behind the last sample and carries no cancellation residue.  (A "total - prefix" form leaves
    // ~1e-16 |total| there, which the chain rule multiplies by delta * |ray| -- 1e10 for the last interval, up
    // to 1e8 for inverse-depth samples: with a depth loss on LLFF-type rays |total| ~ 1e8 and the residue became
    // a spurious O(100) gradient on the far samples.)
    double carry = 0.0;
    float dlen = 0.f;
    for (int j0 = ((N - 1) / 64) * 64; j0 >= 0; j0 -= 64) {
        const int i = j0 + lane;
        const bool ok = i < N;
        float w = 0.f, q = 0.f, tt = 0.f, dens = 0.f, delta = 0.f, raw = 0.f, Tn1 = 0.f;
        const float* c = a.rgb_samples + (base + (ok ? i : 0)) * 3;
        if (ok) {
            tt = a.t[base + i];
            w = a.weights[base + i];
            q = gC[0] * c[0] + gC[1] * c[1] + gC[2] * c[2] + gD * tt + gO + (gw ? gw[i] : 0.f);
            if (sg.g_depth_var) q += gV * ((tt - Dr) * (tt - Dr) - 2.0f * tt * Dr * (1.0f - Or));
            if (sg.g_rgb_var) q += gU * ((c[0] + c[1] + c[2]) * (1.0f - Or) - SC);
            float tn = i + 1 < N ? a.t[base + i + 1] : 0.f;
            delta = i + 1 < N ? __fsub_rn(tn, tt) : 1e10f;
            raw = a.sigma_raw[base + i];
            if (a.noise && noise_scale != 0.0f) raw = __fadd_rn(raw, __fmul_rn(a.noise[base + i], noise_scale));
            dens = a.direct ? raw : softplus_f(raw);
            Tn1 = a.d_sigma_raw[base + i];
        }
        const double wq = (double)w * (double)q;
        const double rincl = wave_rev_incl_scan(wq, lane);          // sum_{k>=i, same chunk} w_k q_k
        const double rnext = __shfl_down(rincl, 1);
        const double suffix_d = carry + (lane < 63 ? rnext : 0.0);  // sum_{k>i} w_k q_k
        carry += __shfl(rincl, 0);
        if (ok) {
            float suffix = (float)suffix_d;
            float ds = Tn1 * q - suffix;
            if (i < N - 2) ds -= gA * A;
            float dist = delta * ell;
            float dsig = ds * dist + (gdn ? gdn[i] : 0.f);
            float draw = a.direct ? dsig : dsig * (raw > 20.0f ? 1.0f : sigmoid_f(raw));
            a.d_sigma_raw[base + i] = draw;
            dlen += ds * dens * delta;
            float* dz = a.d_z + (base + i) * 3;
            const float wu = w * gU * (1.0f - Or);                  // rgb_var = SC (1 - O): every channel of c_i enters with w_i (1 - O)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float dc = w * gC[ch] + wu + (grs ? grs[i * 3 + ch] : 0.f);
                dz[ch] = a.direct ? dc : dc * c[ch] * (1.0f - c[ch]);
            }
        }
    }
    dlen = wave_sumf(dlen);
    if (lane == 0 && a.d_len) a.d_len[ray] = dlen;
}

// ------------------------------------------------------------------ fine samples
// torch.linspace(start, end, steps)[i] as the CPU kernel computes it (symmetric form)
static SP_DEV float linspace_at(float start, float end, int steps, int i) {
    float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? __fadd_rn(start, __fmul_rn(step, (float)i)) : __fsub_rn(end, __fmul_rn(step, (float)(steps - i - 1)));
}

__global__ void __launch_bounds__(64) sample_fine_kernel(SampleFineArgs a) {
    extern __shared__ float sm[];
    const int ray = blockIdx.x, lane = threadIdx.x;
    const int Nc = a.n_coarse, Nf = a.n_fine, Nt = Nc + Nf;
    float* cdf = sm;                    // [Nc+1]
    float* srt = sm + (Nc + 1);         // [P] sort buffer
    int P = 1;
    while (P < Nt) P <<= 1;
    const float* w = a.weights + (int64_t)ray * Nc;
    const float dmin = a.range_dev ? a.range_dev[0] : a.dmin, dmax = a.range_dev ? a.range_dev[1] : a.dmax;
    double tot = 0.0;
    for (int i = lane; i < Nc; i += 64) tot += (double)w[i];
    const float denom = __fadd_rn((float)wave_sum(tot), 1e-6f);
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.f;
    for (int j0 = 0; j0 < Nc; j0 += 64) {
        const int i = j0 + lane;
        float pdf = i < Nc ? __fdiv_rn(w[i], denom) : 0.f;
        double incl = carry + wave_incl_scan((double)pdf, lane);
        carry = __shfl(incl, 63);
        if (i < Nc) cdf[i + 1] = (float)incl;
    }
    __syncthreads();
    const float* tc = a.t_coarse + (int64_t)ray * Nc;
    for (int i = lane; i < Nc; i += 64) srt[i] = tc[i];
    for (int j = lane; j < Nf; j += 64) {
        const float u = a.u_mid ? a.u_mid[j] : a.u_host[j];
        // idx = #{k in [0,Nc] : cdf[k] <= u}   (searchsorted right=True)
        int lo = 0, hi = Nc + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int il = lo - 1 < 0 ? 0 : lo - 1, ih = lo > Nc ? Nc : lo;
        const float dl = linspace_at(dmin, dmax, Nc + 1, il), dh = linspace_at(dmin, dmax, Nc + 1, ih);
        const float cl = cdf[il], chh = cdf[ih];
        const float frac = __fdiv_rn(__fsub_rn(u, cl), __fadd_rn(__fsub_rn(chh, cl), 1e-8f));
        const float tf = __fadd_rn(dl, __fmul_rn(frac, __fsub_rn(dh, dl)));
        srt[Nc + j] = tf;
        if (a.t_fine) a.t_fine[(int64_t)ray * Nf + j] = tf;
    }
    for (int i = Nt + lane; i < P; i += 64) srt[i] = __builtin_inff();
    __syncthreads();
    // bitonic sort, ascending
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += 64) {
                int p = i ^ j;
                if (p > i) {
                    float x = srt[i], y = srt[p];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { srt[i] = y; srt[p] = x; }
                }
            }
            __syncthreads();
        }
    float* out = a.t_out + (int64_t)ray * Nt;
    for (int i = lane; i < Nt; i += 64) out[i] = srt[i];
}

// ------------------------------------------------------------------ ray gradient reduction
// d_center = sum_s dp ; d_ray = sum_s t dp + (I - dd^T)/|r| * dL/dd + dL/d|r| * r/|r|
// where dL/dd comes from the view-encoding gradient summed over the ray's samples.
__global__ void __launch_bounds__(64) ray_reduce_kernel(RayReduceArgs a) {
    const int ray = a.ray_base + blockIdx.x, lane = threadIdx.x;
    const int N = a.nsamp;
    const int64_t base = (int64_t)ray * N;
    float sc[3] = {0, 0, 0}, sr[3] = {0, 0, 0};
    for (int i = lane; i < N; i += 64) {
        const float* dp = a.dp + (base + i) * 3;
        float tt = a.t[base + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) { sc[c] += dp[c]; sr[c] += tt * dp[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { sc[c] = wave_sumf(sc[c]); sr[c] = wave_sumf(sr[c]); }
    // view-encoding gradient: column sums of dv[row][32] (fp32, pos layout with CH = 4)
    float dvf = 0.f;      // lane p < 32 ends up with column p
    if (a.dv) {
        // lane half h takes samples i = h (mod 2), four independent partial sums each (a single dependent chain
        // of N loads per ray was latency-bound: 70 us per 4096-ray launch), fixed combination order
        const int hh = lane >> 5, col = lane & 31;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = hh;
        for (; i + 6 < N; i += 8) {
            s0 += a.dv[(base + i) * 32 + col];
            s1 += a.dv[(base + i + 2) * 32 + col];
            s2 += a.dv[(base + i + 4) * 32 + col];
            s3 += a.dv[(base + i + 6) * 32 + col];
        }
        for (; i < N; i += 2) s0 += a.dv[(base + i) * 32 + col];
        dvf = (s0 + s1) + (s2 + s3);
        dvf += __shfl_down(dvf, 32);
    }
    float x = a.dir[ray * 3], y = a.dir[ray * 3 + 1], z = a.dir[ray * 3 + 2];
    float len = a.raylen[ray];
    float inv = fmaxf(len, 1e-12f);
    float d[3] = {x / inv, y / inv, z / inv};
    float gd[3] = {0, 0, 0};
    if (a.dv) {
        // lane p holds dL/dv of feature feat = view_feat(slot of column p);
        // features: [d(3), per coord c: sin k0..3, cos k0..3]
        float contrib = 0.f;
        int coord = -1;
        const int feat = lane < 32 ? view_feat(q_of_pos(lane, 4), h_of_pos(lane, 4)) : -1;
        if (feat >= 0 && feat < 3) { coord = feat; contrib = dvf; }
        else if (feat >= 3) {
            int f = feat - 3; coord = f / 8;
            int k = f % 4; bool is_cos = (f % 8) >= 4;
            float s, co;
            sincosf(__fmul_rn(d[coord], pe_freq(k)), &s, &co);
            float m = a.c2f_view[k] * pe_freq(k);
            contrib = dvf * m * (is_cos ? -s : co);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gd[c] = wave_sumf(coord == c ? contrib : 0.f);
    }
    if (lane == 0) {
        float dl = a.d_len ? a.d_len[ray] : 0.f;
        float dot = gd[0] * d[0] + gd[1] * d[1] + gd[2] * d[2];
        float r[3] = {x, y, z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g = sr[c] + dl * r[c] / inv;
            if (len > 1e-12f) g += (gd[c] - dot * d[c]) / inv;
            else g += gd[c] / inv;
            a.d_center[ray * 3 + c] = a.accumulate ? a.d_center[ray * 3 + c] + sc[c] : sc[c];
            a.d_dir[ray * 3 + c] = a.accumulate ? a.d_dir[ray * 3 + c] + g : g;
        }
    }
}

// ------------------------------------------------------------------ launchers
// ------------------------------------------------------------------ far rows: transplant of the saved activations
// The far launch of a pass (sparf_hip.h "far rows") evaluated nrays * K sample rows in fp32 and saved, in its own fp32 tile-block
// area, what a forward saves: the layer inputs and the ReLU mask words.  The BACKWARD of those rows does not need fp32 -- it is linear
// in the upstream gradient, and its operands (weights as head + tail, gradients and saved activations as bf16 heads) are what every
// other row of the pass uses -- it needs the far forward's DECISIONS (masks) and activations.  This kernel writes them where the main
// backward will read them: far row r stands for sample row g = routed_row(r) of the pass; its 2272 saved values go, rounded to bf16,
// to row g % 32 of tile g / 32 of the main (bf16-plane) save area, its mask words to that row's two lane slots.  One workgroup per
// 32-row far tile.  Layouts (layout.h): a buffer is [16-byte chunk][row & 31][CH elements] in `pos` order, CH = 4 (fp32) / 8 (bf16):
// bf16 chunk 2a + h = slots q = 8a .. 8a+7 of half h = fp32 chunks 4a + h (q % 8 < 4) and 4a + 2 + h.
__global__ void __launch_bounds__(256) far_transplant_kernel(const char* __restrict__ far_area, char* __restrict__ main_area, int64_t frows,
                                                             int K, int nsamp, int64_t main_tile_bytes, int main_mask_off) {
    constexpr int64_t FT = save_tile_bytes(PREC_FP32);
    constexpr int FMASK = save_mask_tile_off(PREC_FP32, 0);
    constexpr int NCH8 = SAVE_COLS / 8;                      // 16-byte bf16 chunks per row over all buffers (buffer widths are multiples of 32)
    const int64_t ft = blockIdx.x;
    const char* src_tile = far_area + ft * FT;
    __shared__ int64_t dst_row_off[32];                      // byte offset of (tile, row) of each far row inside the main area, -1: past the end
    if (threadIdx.x < 32) {
        const int64_t r = ft * 32 + threadIdx.x;
        int64_t off = -1;
        if (r < frows) {
            const int64_t g = (r / K) * nsamp + (nsamp - K) + r % K;
            off = (g >> 5) * main_tile_bytes + (g & 31) * 16;
        }
        dst_row_off[threadIdx.x] = off;
    }
    __syncthreads();
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    for (int i = threadIdx.x; i < 32 * NCH8; i += blockDim.x) {
        const int j = i & 31, c8g = i >> 5;                  // far row, bf16 chunk counted over the whole saved row
        const int64_t drow = dst_row_off[j];
        if (drow < 0) continue;
        // chunk c8g lies in the buffer whose chunk range holds it: buffers start at save_coloff(b) / 8 chunks
        int b = 0, c0 = 0;
#pragma unroll
        for (int bb = 0; bb < SB_COUNT; ++bb) {
            const int start = (int)(save_coloff(bb) / 8);
            if (c8g >= start) { b = bb; c0 = start; }
        }
        const int c8 = c8g - c0, a = c8 >> 1, h = c8 & 1;
        const char* sb = src_tile + (int64_t)save_coloff(b) * 32 * 4;         // fp32 buffer inside the far tile
        const f32x4 lo = *(const f32x4*)(sb + (int64_t)(4 * a + h) * 512 + j * 16);
        const f32x4 hi = *(const f32x4*)(sb + (int64_t)(4 * a + 2 + h) * 512 + j * 16);
        u32x4 out;
        { const bf16x2_t v = {(__bf16)lo[0], (__bf16)lo[1]}; out[0] = __builtin_bit_cast(unsigned, v); }
        { const bf16x2_t v = {(__bf16)lo[2], (__bf16)lo[3]}; out[1] = __builtin_bit_cast(unsigned, v); }
        { const bf16x2_t v = {(__bf16)hi[0], (__bf16)hi[1]}; out[2] = __builtin_bit_cast(unsigned, v); }
        { const bf16x2_t v = {(__bf16)hi[2], (__bf16)hi[3]}; out[3] = __builtin_bit_cast(unsigned, v); }
        *(u32x4*)(main_area + drow + (int64_t)save_coloff(b) * 32 * 2 + (int64_t)c8 * 512) = out;
    }
    // mask words: lane (n, h) of the far tile -> lane slot (g % 32 + 32 h) of the main tile's mask block of the same buffer
    for (int i = threadIdx.x; i < 64 * SB_COUNT; i += blockDim.x) {
        const int lane = i & 63, b = i >> 6, j = lane & 31, h = lane >> 5;
        const int64_t drow = dst_row_off[j];
        if (drow < 0) continue;
        const u32x4 w = *(const u32x4*)(src_tile + FMASK + b * MASK_TILE_BYTES + lane * 16);
        // drow = tile * tile_bytes + row * 16: the row's lane slot of half h sits 32 lanes = 512 B further
        *(u32x4*)(main_area + drow + main_mask_off + b * MASK_TILE_BYTES + h * 512) = w;
    }
}

int launch_far_transplant(int main_prec, const void* far_area, void* main_area, int64_t frows, int far_count, int nsamp, hipStream_t s) {
    if (frows <= 0) return 0;
    if (main_prec == PREC_FP32 || nplanes_of(main_prec) != 1) return 1;       // bf16-plane save areas only (bf16, bf16x3 with head planes)
    const int64_t ntiles = (frows + 31) / 32;
    hipLaunchKernelGGL(far_transplant_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, (const char*)far_area, (char*)main_area, frows, far_count, nsamp,
                       save_tile_bytes(main_prec), save_mask_tile_off(main_prec, 0));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

__global__ void ray_len_kernel(const float* __restrict__ dir, int nrays, float* __restrict__ raylen) {
    int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= nrays) return;
    float x = dir[ray * 3], y = dir[ray * 3 + 1], z = dir[ray * 3 + 2];
    raylen[ray] = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}
__global__ void len_to_dir_kernel(const float* __restrict__ dir, const float* __restrict__ raylen, const float* __restrict__ d_len, int nrays,
                                  float* __restrict__ d_dir) {
    int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= nrays) return;
    const float inv = fmaxf(raylen[ray], 1e-12f), g = d_len[ray];
#pragma unroll
    for (int c = 0; c < 3; ++c) d_dir[ray * 3 + c] = g * dir[ray * 3 + c] / inv;
}
int launch_len_to_dir(const float* dir, const float* raylen, const float* d_len, int nrays, float* d_dir, hipStream_t s) {
    if (nrays <= 0) return 0;
    hipLaunchKernelGGL(len_to_dir_kernel, dim3((nrays + 255) / 256), dim3(256), 0, s, dir, raylen, d_len, nrays, d_dir);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_ray_setup(int prec, const float* dir, int nrays, const float* c2f_view, void* venc, float* raylen, hipStream_t s) {
    if (nrays <= 0) return 0;
    dim3 g((nrays + 255) / 256), b(256);
    if (prec < 0) {
        hipLaunchKernelGGL(ray_len_kernel, g, b, 0, s, dir, nrays, raylen);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (prec == PREC_BF16) hipLaunchKernelGGL(ray_setup_kernel<PREC_BF16>, g, b, 0, s, dir, nrays, c2f_view, (__bf16*)venc, raylen);
    else if (prec == PREC_FP32) hipLaunchKernelGGL(ray_setup_kernel<PREC_FP32>, g, b, 0, s, dir, nrays, c2f_view, (float*)venc, raylen);
    else if (prec == PREC_X3) hipLaunchKernelGGL(ray_setup_kernel<PREC_X3>, g, b, 0, s, dir, nrays, c2f_view, (float*)venc, raylen);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_sample_coarse(const float* jitter, float u_const, const float* dmax_ray, const float* range_dev, float dmin, float scale,
                         int inverse, int64_t rows, int nsamp, float* t, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, jitter, u_const, dmax_ray, range_dev,
                       dmin, scale, inverse, rows, nsamp, t);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_composite_fwd(const CompositeFwdArgs& a, hipStream_t s) {
    if (a.nrays <= 0) return 0;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(a.nrays), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_composite_bwd(const CompositeBwdArgs& a, hipStream_t s) {
    if (a.nrays <= 0) return 0;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(a.nrays), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_sample_fine(const SampleFineArgs& a, hipStream_t s) {
    if (a.nrays <= 0) return 0;
    int P = 1;
    while (P < a.n_coarse + a.n_fine) P <<= 1;
    size_t smem = (size_t)(a.n_coarse + 1 + P) * sizeof(float);
    if (smem > 64 * 1024) return 3;
    hipLaunchKernelGGL(sample_fine_kernel, dim3(a.nrays), dim3(64), smem, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_ray_reduce(const RayReduceArgs& a, hipStream_t s) {
    if (a.nrays <= 0) return 0;
    hipLaunchKernelGGL(ray_reduce_kernel, dim3(a.nrays), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}


// ------------------------------------------------------------------ ray generation
// camera.get_center_and_ray / get_center_and_ray_at_pixels
// (/root/reference/source/utils/camera.py:347-416 with img2cam :296-306, cam2world :321-326,
// Pose.invert): for pixel (x, y) of image b
//     g      = K_b^-1 [x, y, 1]            (camera frame, z = 1: the ray is NOT normalised)
//     R_inv  = R_b^T,  t_inv = -(R_inv t_b)             (inverse of the w2c pose [R|t])
//     center = t_inv,  ray = (R_inv g + t_inv) - t_inv   (the reference subtracts two world
//                                                        points; the same order is kept here)
// Flat indices address pixel centres (x + 0.5, y + 0.5), explicit pixels are used as given.
// K^-1 is formed per thread from the adjugate in double precision and rounded once.
struct RayGeom { float ki[9], ri[9], ti[3]; };

static SP_DEV RayGeom ray_geom(const float* pose, const float* intr, int b) {
    RayGeom g;
    const float* K = intr + b * 9;
    const double a = K[0], bb = K[1], c = K[2], d = K[3], e = K[4], f = K[5], gg = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * gg), C = d * h - e * gg;
    const double det = a * A + bb * B + c * C, inv = 1.0 / det;
    g.ki[0] = (float)(A * inv); g.ki[1] = (float)(-(bb * i - c * h) * inv); g.ki[2] = (float)((bb * f - c * e) * inv);
    g.ki[3] = (float)(B * inv); g.ki[4] = (float)((a * i - c * gg) * inv);  g.ki[5] = (float)(-(a * f - c * d) * inv);
    g.ki[6] = (float)(C * inv); g.ki[7] = (float)(-(a * h - bb * gg) * inv); g.ki[8] = (float)((a * e - bb * d) * inv);
    const float* P = pose + b * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) g.ri[r * 3 + q] = P[q * 4 + r];          // R^T
#pragma unroll
    for (int r = 0; r < 3; ++r) g.ti[r] = -(g.ri[r * 3] * P[3] + g.ri[r * 3 + 1] * P[7] + g.ri[r * 3 + 2] * P[11]);
    return g;
}

static SP_DEV void ray_pixel(const RayGenArgs& a, int b, int r, float& x, float& y) {
    const int64_t src = (a.per_image ? (int64_t)b * a.nrays : 0) + r;
    if (a.pixels) {
        x = a.pixels[src * 2];
        y = a.pixels[src * 2 + 1];
    } else {
        const int64_t idx = a.ray_idx[src];
        x = (float)(idx % a.width) + 0.5f;
        y = (float)(idx / a.width) + 0.5f;
    }
}

__global__ void ray_gen_fwd_kernel(RayGenArgs a) {
    const int b = blockIdx.y;
    const RayGeom g = ray_geom(a.pose, a.intr, b);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.nrays; r += gridDim.x * blockDim.x) {
        float x, y;
        ray_pixel(a, b, r, x, y);
        float gc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) gc[i] = x * g.ki[i * 3] + y * g.ki[i * 3 + 1] + g.ki[i * 3 + 2];
        const int64_t o = ((int64_t)b * a.nrays + r) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float w = gc[0] * g.ri[i * 3] + gc[1] * g.ri[i * 3 + 1] + gc[2] * g.ri[i * 3 + 2] + g.ti[i];
            a.center[o + i] = g.ti[i];
            a.ray[o + i] = w - g.ti[i];
        }
    }
}

// d pose[b] (w2c [R|t]) from d center, d ray:  ray_i = sum_j R_ji g_j,  center_i = -sum_j R_ji t_j
//   dR_ji = sum_rays (g_j d_ray_i - t_j d_center_i),   dt_j = -sum_rays sum_i R_ji d_center_i
// one workgroup per image, fixed-order tree reduction (deterministic)
__global__ void __launch_bounds__(256) ray_gen_bwd_kernel(RayGenArgs a, const float* d_center, const float* d_ray, float* d_pose) {
    const int b = blockIdx.x;
    const RayGeom g = ray_geom(a.pose, a.intr, b);
    const float* P = a.pose + b * 12;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < a.nrays; r += blockDim.x) {
        float x, y;
        ray_pixel(a, b, r, x, y);
        float gc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) gc[i] = x * g.ki[i * 3] + y * g.ki[i * 3 + 1] + g.ki[i * 3 + 2];
        const int64_t o = ((int64_t)b * a.nrays + r) * 3;
        float dr[3] = {0.f, 0.f, 0.f}, dc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (d_ray) dr[i] = d_ray[o + i];
            if (d_center) dc[i] = d_center[o + i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[j * 4 + i] += gc[j] * dr[i] - P[j * 4 + 3] * dc[i];
            acc[j * 4 + 3] -= P[j * 4] * dc[0] + P[j * 4 + 1] * dc[1] + P[j * 4 + 2] * dc[2];
        }
    }
    __shared__ float red[256][13];
#pragma unroll
    for (int k = 0; k < 12; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 12; ++k) red[threadIdx.x][k] += red[threadIdx.x + s][k];
        __syncthreads();
    }
    if (threadIdx.x < 12) d_pose[b * 12 + threadIdx.x] = red[0][threadIdx.x];
}

int launch_ray_gen_fwd(const RayGenArgs& a, hipStream_t s) {
    if (a.nimg <= 0 || a.nrays <= 0) return 0;
    int gx = (a.nrays + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ray_gen_fwd_kernel, dim3(gx, a.nimg), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_ray_gen_bwd(const RayGenArgs& a, const float* d_center, const float* d_ray, float* d_pose, hipStream_t s) {
    if (a.nimg <= 0) return 0;
    hipLaunchKernelGGL(ray_gen_bwd_kernel, dim3(a.nimg), dim3(256), 0, s, a, d_center, d_ray, d_pose);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
