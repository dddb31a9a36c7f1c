// Weight gradients: dW_l[out][in] = sum over sample rows of dY_l[row][out] * X_l[row][in]
// for the ten layers, as eleven split-K MFMA GEMMs ("jobs", layout.h) over the activations
// saved by the forward kernel (X) and the pre-activation gradients written by the dgrad
// kernel (dY).  The contraction runs over rows, so both operands are transposed on their
// way LDS -> registers.  Each workgroup owns one (job, row-range) pair and writes an fp32
// partial [32*MB][32*NB] matrix (+ the bias gradient = column sums of dY); a second kernel
// sums the partials over the splits and scatters them into nn.Linear [out][in] order.
// Deterministic: no atomics.
//
// HBM-bound by design: 2*(M+N)*sizeof(act) bytes per row against 2*M*N flops.
#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

enum { WG_ROWS = 32, WG_THREADS = 512 };

template <int PREC> struct WOps;
template <> struct WOps<PREC_BF16> {
    typedef Policy<PREC_BF16> P;
    enum { KSTEPS = WG_ROWS / 16, UNROLL = 2 };
    // 8 consecutive rows of one column: rows r0..r0+7, column col
    static SP_DEV bf16x8 frag(const __bf16* tile, int stride, int kk, int lane, int col0) {
        const __bf16* p = tile + (kk * 16 + (lane >> 5) * 8) * stride + col0 + (lane & 31);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j * stride];
        return v;
    }
    static SP_DEV float fsum(bf16x8 v) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j];
        return s;
    }
};
template <> struct WOps<PREC_FP32> {
    typedef Policy<PREC_FP32> P;
    enum { KSTEPS = WG_ROWS / 2, UNROLL = 1 };
    static SP_DEV float frag(const float* tile, int stride, int kk, int lane, int col0) {
        return tile[(kk * 2 + (lane >> 5)) * stride + col0 + (lane & 31)];
    }
    static SP_DEV float fsum(float v) { return v; }
};

template <int PREC, int MB, int NB>
SP_DEV void wgrad_job(const WgradArgs& a, int job, char* lds) {
    typedef Policy<PREC> P;
    typedef typename P::act_t act_t;
    typedef WOps<PREC> W;
    constexpr int AB = (int)sizeof(act_t);
    constexpr int M = 32 * MB, N = 32 * NB;
    constexpr int NBW = (NB + 7) / 8;                         // n-blocks per wave
    constexpr int EPV = 16 / AB;                              // elements per 16-byte piece
    constexpr int PIECES = WG_ROWS * (M + N) / EPV;
    constexpr int NPT = (PIECES + WG_THREADS - 1) / WG_THREADS;

    const WJob jb = wjob(job);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gcols = grad_cols(jb.gbuf), scols = save_cols(jb.sbuf);
    const act_t* dy_base = (const act_t*)a.grad + a.rows * grad_coloff(jb.gbuf);
    const act_t* x_base = (const act_t*)a.save + a.rows * save_coloff(jb.sbuf) + jb.xcol0;

    act_t* dy_t = (act_t*)lds;               // [WG_ROWS][M]
    act_t* x_t = dy_t + WG_ROWS * M;         // [WG_ROWS][N]

    const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_split;
    const int64_t r_end = r_begin + a.rows_per_split < a.rows ? r_begin + a.rows_per_split : a.rows;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NBW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
    float bsum[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) bsum[m] = 0.f;

    u32x4 stage[NPT];
    auto load_tile = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int p = threadIdx.x + i * WG_THREADS;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p < PIECES) {
                const bool is_x = p >= WG_ROWS * M / EPV;
                const int pp = is_x ? p - WG_ROWS * M / EPV : p;
                const int per_row = (is_x ? N : M) / EPV;
                const int row = pp / per_row, c16 = pp % per_row;
                const int64_t grow = r0 + row;
                if (grow < r_end) {
                    const act_t* src = is_x ? x_base + grow * scols + c16 * EPV : dy_base + grow * gcols + c16 * EPV;
                    v = *(const u32x4*)src;
                }
            }
            stage[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int p = threadIdx.x + i * WG_THREADS;
            if (p < PIECES) *(u32x4*)(lds + (int64_t)p * 16) = stage[i];   // tiles are contiguous: piece p -> byte 16p
        }
    };

    if (r_begin < r_end) load_tile(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
        store_tile();
        __syncthreads();
        if (r0 + WG_ROWS < r_end) load_tile(r0 + WG_ROWS);
#pragma unroll W::UNROLL
        for (int kk = 0; kk < W::KSTEPS; ++kk) {
            typename P::B bfr[NBW];
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                const int nb = wave + 8 * i;
                bfr[i] = nb < NB ? W::frag(x_t, N, kk, lane, nb * 32) : P::zero();
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                typename P::A afr = W::frag(dy_t, M, kk, lane, m * 32);
                if (wave == 0) bsum[m] += W::fsum(afr);
#pragma unroll
                for (int i = 0; i < NBW; ++i)
                    if (wave + 8 * i < NB) acc[m][i] = P::mfma(afr, bfr[i], acc[m][i]);
            }
        }
        __syncthreads();
    }

    float* out = a.partial + (int64_t)blockIdx.x * wpartial_floats();
    float* mat = out + wjob_mat_off(job);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = wave + 8 * i;
        if (nb < NB) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int po = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                    mat[(int64_t)po * N + nb * 32 + (lane & 31)] = acc[m][i][r];
                }
        }
    }
    if (wave == 0) {
        float* bo = out + wjob_bias_off(job);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float s = bsum[m] + __shfl_xor(bsum[m], 32);
            if (lane < 32) bo[32 * m + lane] = s;
        }
    }
}

template <int PREC>
__global__ void __launch_bounds__(WG_THREADS) wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int job = blockIdx.y;
    switch (job) {
        case 0: case 5: wgrad_job<PREC, 8, 2>(a, job, lds); break;
        case 8: wgrad_job<PREC, 9, 8>(a, job, lds); break;
        case 9: wgrad_job<PREC, 4, 9>(a, job, lds); break;
        case 10: wgrad_job<PREC, 1, 4>(a, job, lds); break;
        default: wgrad_job<PREC, 8, 8>(a, job, lds); break;
    }
}

// out[p] = sum_s partial[s][wsrc[p]]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, const int32_t* __restrict__ wsrc,
                                    float* __restrict__ out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N_PARAMS) return;
    const int64_t src = wsrc[p];
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * wpartial_floats() + src];
    out[p] = s;
}

int launch_wgrad(int prec, const WgradArgs& a, int nsplit, const int32_t* wsrc, float* grad_out, hipStream_t s) {
    if (a.rows <= 0 || nsplit <= 0) return 1;
    const int ab = abytes_of(prec);
    const size_t smem = (size_t)WG_ROWS * (288 + 256) * ab;
    dim3 grid(nsplit, N_WJOBS), block(WG_THREADS);
    if (prec == PREC_BF16) hipLaunchKernelGGL(wgrad_kernel<PREC_BF16>, grid, block, smem, s, a);
    else if (prec == PREC_FP32) {
        hipFuncSetAttribute((const void*)wgrad_kernel<PREC_FP32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(wgrad_kernel<PREC_FP32>, grid, block, smem, s, a);
    } else return 1;
    if (hipGetLastError() != hipSuccess) return 2;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((N_PARAMS + 255) / 256), dim3(256), 0, s, a.partial, nsplit, wsrc, grad_out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
