// Weight packing: nn.Linear fp32 parameters -> the MFMA fragment streams of streams.h
// (forward A = W, backward A = W^T) and packed accumulator-initial biases: pure gathers driven
// by the host-built tables of tables.cpp.  c2f_kernel: the BARF coarse-to-fine band weights
// from the device-resident `progress` scalar (no host sync), one 16-float vector per pass.
#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

struct ParamPtrs { const float* p[2 * N_LAYERS]; };   // W0, b0, W1, b1, ...

// flat parameter index -> value.  Pointer table and layer offsets sit in LDS: a by-value
// kernel-argument array indexed with a runtime value would be demoted to scratch memory.
struct ParamLut {
    const float* ptr[2 * N_LAYERS];
    int off[2 * N_LAYERS + 1];     // start of W0, b0, W1, b1, ..., end
};
template <int L> struct LayerOff { enum : int { W = (int)param_w_off(L), B = (int)param_b_off(L) }; };
static SP_DEV void constexpr_store(int* off, int l) {
    // all ten (w, b) offsets as immediates
    static_for<N_LAYERS>([&](auto lc) {
        constexpr int li = decltype(lc)::value;
        if (l == li) { off[2 * li] = LayerOff<li>::W; off[2 * li + 1] = LayerOff<li>::B; }
    });
}
static SP_DEV float param_at(const ParamLut& lut, int idx) {
    int lo = 0, hi = 2 * N_LAYERS;               // largest t with off[t] <= idx
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        if (lut.off[mid] <= idx) lo = mid; else hi = mid - 1;
    }
    return lut.ptr[lo][idx - lut.off[lo]];
}

template <int PREC>
__global__ void __launch_bounds__(256) pack_kernel(ParamPtrs pp, const int32_t* __restrict__ tables, char* __restrict__ out) {
    typedef typename Policy<PREC>::act_t act_t;
    constexpr int64_t NSTREAM = (fwd_stream_bytes(PREC) + bwd_stream_bytes(PREC)) / abytes_of(PREC);      // logical elements
    constexpr int64_t NTOT = NSTREAM + AUX_PK_FLOATS;          // + packed biases + raw-coordinate columns
    // forced compile-time: left as plain calls these layout functions become run-time loops
    constexpr int64_t TBL_BIAS = tbl_bias_off(PREC), OUT_BIAS = packed_bias_off(PREC);
    __shared__ ParamLut lut;
#pragma unroll
    for (int i = 0; i < 2 * N_LAYERS; ++i)
        if (threadIdx.x == i) lut.ptr[i] = pp.p[i];        // static indices: kernel args stay in SGPRs
    if (threadIdx.x <= N_LAYERS) {
        if (threadIdx.x < N_LAYERS) {
#pragma unroll
            for (int l = 0; l < N_LAYERS; ++l)
                if (threadIdx.x == l) {
                    constexpr_store(lut.off, l);
                }
        } else {
            lut.off[2 * N_LAYERS] = N_PARAMS;
        }
    }
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < NTOT; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < NSTREAM) {
            // tables: [fwd elements][bwd elements] are contiguous, as are the two streams in `out`
            const int idx = tables[e];
            const float w = idx < 0 ? 0.0f : param_at(lut, idx);
            if constexpr (PREC == PREC_X3) {
                // fragment f = e / 512 occupies [f * 2 KiB, +1 KiB) heads and [+1 KiB, +2 KiB) tails
                const int64_t f = e >> 9, i = e & 511;
                const __bf16 hi = (__bf16)w;
                ((__bf16*)(out + f * 2048))[i] = hi;
                ((__bf16*)(out + f * 2048 + 1024))[i] = (__bf16)(w - (float)hi);
            } else {
                ((act_t*)out)[e] = (act_t)w;
            }
        } else {
            const int idx = tables[TBL_BIAS + (e - NSTREAM)];
            ((float*)(out + OUT_BIAS))[e - NSTREAM] = idx < 0 ? 0.0f : param_at(lut, idx);
        }
    }
}

// band weights: k < 10 -> point encoding (L=10), 10..13 -> view encoding (L=4), 14..15 pad
// frequency_nerf.py:248-253: w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2, alpha = (progress - start) / (end - start) * L
__global__ void c2f_kernel(const float* __restrict__ progress, int has_c2f, float c2f_start, float c2f_range, float* __restrict__ out) {
    const int j = threadIdx.x;
    if (j >= C2F_FLOATS) return;
    float w = j < 14 ? 1.0f : 0.0f;
    if (has_c2f && j < 14) {
        const int L = j < 10 ? L3D : LVIEW, k = j < 10 ? j : j - 10;
        float alpha = __fmul_rn(__fdiv_rn(__fsub_rn(progress[0], c2f_start), c2f_range), (float)L);
        float x = fminf(fmaxf(__fsub_rn(alpha, (float)k), 0.0f), 1.0f);
        w = __fdiv_rn(__fsub_rn(1.0f, cosf(__fmul_rn(x, 3.14159274101257324219f))), 2.0f);
    }
    out[j] = w;
}

int launch_c2f(const float* progress, int has_c2f, float c2f_start, float c2f_end, float* out, hipStream_t s) {
    const float range = (float)((double)c2f_end - (double)c2f_start);
    hipLaunchKernelGGL(c2f_kernel, dim3(1), dim3(64), 0, s, progress, has_c2f, c2f_start, range, out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_pack(int prec, const float* const* param_ptrs_host, const int32_t* tables, void* out, hipStream_t s) {
    ParamPtrs pp;
    for (int i = 0; i < 2 * N_LAYERS; ++i) pp.p[i] = param_ptrs_host[i];
    if (prec == PREC_BF16)
        hipLaunchKernelGGL(pack_kernel<PREC_BF16>, dim3(2048), dim3(256), 0, s, pp, tables, (char*)out);
    else if (prec == PREC_FP32)
        hipLaunchKernelGGL(pack_kernel<PREC_FP32>, dim3(2048), dim3(256), 0, s, pp, tables, (char*)out);
    else if (prec == PREC_X3)
        hipLaunchKernelGGL(pack_kernel<PREC_X3>, dim3(2048), dim3(256), 0, s, pp, tables, (char*)out);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
