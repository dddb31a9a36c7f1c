// Per-CU throughput of the two vector-memory instructions the fused MLP kernels live on, measured the way
// they use them: one workgroup per CU, W waves, every wave issuing a run of
//   (a) LDS-DMA weight pieces   buffer_load_dwordx4 ... lds   (1 KiB per wave-instruction, L2-resident 1 MiB stream)
//   (b) activation saves        buffer_store_dwordx4 [nt]     (1 KiB contiguous per wave-instruction, HBM-bound stream)
//   (c) both interleaved 1:1
// optionally with M back-to-back MFMAs between two memory instructions.  Reports cycles per memory
// instruction per CU (s_memtime of wave 0, whole run / total instructions of the workgroup) and the byte rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/vmem_probe.hip -o tools/probes/vmem_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int MFMAS, int AUX>
__global__ void __launch_bounds__(512) probe(const char* __restrict__ wsrc, char* __restrict__ dst, int iters, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 1u << 20, 0x00020000);
    char* mydst = dst + ((size_t)blockIdx.x * nw + wave) * (size_t)iters * 1024;
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)mydst, 0, (unsigned)iters * 1024u, 0x00020000);
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((it & 7) * 8 + wave) * 1024), 16, lane * 16,
                                                     ((it * nw + wave) * 1024) & ((1 << 20) - 1), 0, 0);
        if (MODE == 1 || MODE == 2)
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, lane * 16, it * 1024, AUX);
#pragma unroll
        for (int m = 0; m < MFMAS; ++m) {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (sink && acc0[0] + acc1[0] == 12345.678f) sink[0] = acc0[1] + ((float*)lds)[lane];
}

template <int MODE, int MFMAS, int AUX>
void run(const char* name, int waves, const char* wsrc, char* dst, unsigned long long* dcyc, int ncu) {
    const int iters = 512;
    hipLaunchKernelGGL((probe<MODE, MFMAS, AUX>), dim3(ncu), dim3(waves * 64), 0, 0, wsrc, dst, iters, dcyc, (float*)nullptr);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, MFMAS, AUX>), dim3(ncu), dim3(waves * 64), 0, 0, wsrc, dst, iters, dcyc, (float*)nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(ncu);
    hipMemcpy(c.data(), dcyc, ncu * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += (double)x;
    avg /= ncu;
    const int per_it = (MODE == 2 ? 2 : 1);
    const double ninstr = (double)iters * waves * per_it;
    printf("%-34s waves %d mfma/it %d: %7.1f cyc per mem-instr per CU, %6.1f cyc per iteration per wave, %6.1f B/clk/CU, kernel %.3f ms (%.2f TB/s chip)\n", name, waves,
           MFMAS, avg / ninstr, avg / iters, ninstr * 1024.0 / avg, ms, ninstr * 1024.0 * ncu / (ms * 1e-3) / 1e12);
}

int main() {
    int ncu = 256;
    char *wsrc, *dst;
    unsigned long long* dcyc;
    hipMalloc(&wsrc, 1 << 20);
    hipMemset(wsrc, 1, 1 << 20);
    hipMalloc(&dst, (size_t)ncu * 8 * 512 * 1024);
    hipMalloc(&dcyc, ncu * 8);
    for (int waves : {1, 4, 8}) {
        run<0, 0, 0>("lds-dma only", waves, wsrc, dst, dcyc, ncu);
        run<1, 0, 0>("store only (default policy)", waves, wsrc, dst, dcyc, ncu);
        run<1, 0, 2>("store only (nt)", waves, wsrc, dst, dcyc, ncu);
        run<2, 0, 2>("dma + store(nt) 1:1", waves, wsrc, dst, dcyc, ncu);
        run<0, 4, 0>("lds-dma + 4 mfma", waves, wsrc, dst, dcyc, ncu);
        run<1, 4, 2>("store(nt) + 4 mfma", waves, wsrc, dst, dcyc, ncu);
        run<2, 8, 2>("dma + store(nt) + 8 mfma", waves, wsrc, dst, dcyc, ncu);
        run<2, 4, 2>("dma + store(nt) + 4 mfma", waves, wsrc, dst, dcyc, ncu);
        run<0, 8, 0>("lds-dma + 8 mfma", waves, wsrc, dst, dcyc, ncu);
    }
    return 0;
}
