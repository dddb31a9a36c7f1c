// Calibration kernels (measurement only; include/sparf_hip.h "calibration").  Two fixed workloads that know nothing of the
// renderer, timed by bench.py before and after its measurement so that a slow BOX and a slow BUILD can be told apart on the bench
// line: the renderer's kernels change between rounds, these two do not.
//   calib_mfma_kernel : back-to-back v_mfma_f32_32x32x16_bf16 on operands with random bits (two waves per SIMD, four independent
//                       accumulator chains each) -- what the socket sustains under its power cap, in issued bf16 TFLOP/s
//   calib_hbm_kernel  : a read stream through LDS-DMA (global_load_lds_dwordx4 nt into a ring of LDS buffers, the weight-gradient
//                       kernel's operand path) or a register copy (16-byte loads, non-temporal 16-byte stores)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sparf {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cal_bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float cal_f32x16;
typedef unsigned cal_u32x4 __attribute__((ext_vector_type(4)));

static __device__ inline unsigned cal_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// a bf16 pair with random sign / mantissa bits and exponents around 2^-1: |x| in [0.25, 1)
static __device__ inline unsigned cal_pair(unsigned h) {
    const unsigned lo = (h & 0x80ffu) | 0x3e80u | ((h >> 3) & 0x0080u);
    const unsigned hi = ((h >> 16) & 0x80ffu) | 0x3e80u | ((h >> 19) & 0x0080u);
    return lo | (hi << 16);
}

enum { CAL_MFMA_PER_ITER = 16, CAL_MFMA_THREADS = 512 };

__global__ __launch_bounds__(CAL_MFMA_THREADS) void calib_mfma_kernel(int iters, float* sink) {
    const unsigned tid = blockIdx.x * CAL_MFMA_THREADS + threadIdx.x;
    union { cal_bf16x8 v; unsigned u[4]; } a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[i].u[k] = cal_pair(cal_hash(tid * 64u + i * 4 + k));
            b[i].u[k] = cal_pair(cal_hash(tid * 64u + 32 + i * 4 + k));
        }
    cal_f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
        // 4 x 4 operand pairs, accumulator chain j reused every fourth MFMA (8 passes each: no dependent-issue stall)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + j) & 3].v, b[i].v, acc[j], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[tid] = s;          // never true for these operands: keeps the chains alive
}

enum { CAL_HBM_THREADS = 512, CAL_HBM_WAVES = CAL_HBM_THREADS / 64, CAL_HBM_DEPTH = 8, CAL_HBM_PIECE = 1024 };

// mode 0: every wave streams its share of [src, src + bytes) into a ring of CAL_HBM_DEPTH 1 KiB LDS slots (LDS-DMA, counted waits);
// mode 1: 16-byte loads, non-temporal 16-byte stores to dst.  bytes: a multiple of grid * 8 waves * 1 KiB.
__global__ __launch_bounds__(CAL_HBM_THREADS) void calib_hbm_kernel(const char* __restrict__ src, char* __restrict__ dst, int64_t bytes, int mode,
                                                                     float* sink) {
    __shared__ __attribute__((aligned(16))) char ring[CAL_HBM_WAVES * CAL_HBM_DEPTH * CAL_HBM_PIECE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t npieces = bytes / CAL_HBM_PIECE;
    const int64_t stride = (int64_t)gridDim.x * CAL_HBM_WAVES;
    int64_t p = (int64_t)blockIdx.x * CAL_HBM_WAVES + wave;
    if (mode == 0) {
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(ring + wave * CAL_HBM_DEPTH * CAL_HBM_PIECE);
        int slot = 0;
        for (; p < npieces; p += stride) {
            const char* s = src + p * CAL_HBM_PIECE + lane * 16;
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(base + slot * CAL_HBM_PIECE);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(s), "s"(lds_dst) : "memory");
            slot = (slot + 1) & (CAL_HBM_DEPTH - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CAL_HBM_DEPTH - 1) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ring[threadIdx.x] == 0x5a && ring[threadIdx.x + 1] == 0x3c && bytes < 0) sink[threadIdx.x] = 1.0f;
    } else {
        for (; p + 3 * stride < npieces; p += 4 * stride) {
            cal_u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const cal_u32x4*)(src + (p + u * stride) * CAL_HBM_PIECE + lane * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], (cal_u32x4*)(dst + (p + u * stride) * CAL_HBM_PIECE + lane * 16));
        }
        for (; p < npieces; p += stride) {
            const cal_u32x4 v = *(const cal_u32x4*)(src + p * CAL_HBM_PIECE + lane * 16);
            __builtin_nontemporal_store(v, (cal_u32x4*)(dst + p * CAL_HBM_PIECE + lane * 16));
        }
    }
}

int launch_calib_mfma(int iters, float* sink, int grid, hipStream_t s) {
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(grid), dim3(CAL_MFMA_THREADS), 0, s, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_calib_hbm(const void* src, void* dst, int64_t bytes, int mode, float* sink, int grid, hipStream_t s) {
    hipLaunchKernelGGL(calib_hbm_kernel, dim3(grid), dim3(CAL_HBM_THREADS), 0, s, (const char*)src, (char*)dst, bytes, mode, sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
// issued bf16 flops of one calib_mfma launch
int64_t calib_mfma_flops(int iters, int grid) {
    return (int64_t)grid * (CAL_MFMA_THREADS / 64) * iters * CAL_MFMA_PER_ITER * (2LL * 32 * 32 * 16);
}

}  // namespace sparf
