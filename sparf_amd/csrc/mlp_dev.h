// Device-side building blocks of the fused MLP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "streams.h"

namespace sparf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SP_DEV __device__ __forceinline__

template <int N, class F, int... I>
SP_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> SP_DEV void static_for(F&& f) {
    static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------ precision policies
// P::B      B-operand fragment held by a lane for one k-step
// P::NB(w)  number of k-steps of a w-wide vector
template <int PREC> struct Policy;

template <> struct Policy<PREC_BF16> {
    enum { PREC = PREC_BF16, KJ = 8, CH = 8, FRAG_BYTES = 1024, LANE_BYTES = 16, G = group_g(PREC_BF16), NWAVES = nwaves_of(PREC_BF16), PREFETCH = 4 };
    typedef bf16x8 B;
    typedef bf16x8 A;
    typedef __bf16 act_t;
    static SP_DEV B zero() { B z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.0f; return z; }
    static SP_DEV A lds_frag(const char* p) { return *(const bf16x8*)p; }
    static SP_DEV f32x16 mfma(A a, B b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static SP_DEV void set(B* v, int q, float x) { v[q >> 3][q & 7] = (__bf16)x; }
    static SP_DEV float get(const B* v, int q) { return (float)v[q >> 3][q & 7]; }
};

template <> struct Policy<PREC_FP32> {
    enum { PREC = PREC_FP32, KJ = 1, CH = 4, FRAG_BYTES = 256, LANE_BYTES = 4, G = group_g(PREC_FP32), NWAVES = nwaves_of(PREC_FP32), PREFETCH = 4 };
    typedef float B;
    typedef float A;
    typedef float act_t;
    static SP_DEV B zero() { return 0.0f; }
    static SP_DEV A lds_frag(const char* p) { return *(const float*)p; }
    static SP_DEV f32x16 mfma(A a, B b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static SP_DEV void set(B* v, int q, float x) { v[q] = x; }
    static SP_DEV float get(const B* v, int q) { return v[q]; }
};

// ------------------------------------------------------------------ LDS weight pipeline
// Two CHUNK_MAX_BYTES buffers.  acquire(next) = "the current chunk has landed and every
// wave has finished with the other buffer; start fetching `next` into it".  Chunks are
// identified by compile-time byte offset/size inside the packed stream.
// Measured alternatives that did NOT pay (same-box A/B, tools/build_variant.py): a ring of
// three buffers with counted s_waitcnt vmcnt(N) so that the barrier does not drain the
// wave's activation stores (equal within noise; the run-time wait ladder cost the inference
// kernel 5 %), and issuing the stores one at a time between MFMAs (forward 1.26 -> 1.71 ms:
// a VMEM instruction among MFMAs costs ~100 issue cycles).  Stores are cheapest in a short
// burst right after a chunk barrier, while the wave waits for its first LDS fragments.
enum { PIPE_LDS_BYTES = 2 * CHUNK_MAX_BYTES };

template <int NWAVES> struct WeightPipe {
    __amdgpu_buffer_rsrc_t rsrc;   // packed stream (global), addressed as raw buffer
    char* lds;                     // 2 * CHUNK_MAX_BYTES
    int wave, lane16;
    unsigned parity;

    SP_DEV void init(const char* g, unsigned stream_bytes, char* l) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, stream_bytes, 0x00020000);
        lds = l;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane16 = (threadIdx.x & 63) * 16;
        parity = 0;
    }
    // issue this wave's share of chunk [off, off+bytes) into buffer `buf`:
    // buffer_load_dwordx4 ... offen lds  (LDS-DMA, 1 KiB per wave-instruction, address
    // = descriptor base + scalar offset + lane*16, destination = M0 + lane*16)
    SP_DEV void fetch(int off, int bytes, unsigned buf) {
        char* dst = lds + buf * CHUNK_MAX_BYTES;
#pragma unroll
        for (int i = 0; i < CHUNK_MAX_BYTES / (NWAVES * 1024); ++i) {
            int o = (i * NWAVES + wave) * 1024;
            if (o < bytes)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + o), 16,
                                                         lane16, off + o, 0, 0);
        }
    }
    SP_DEV void prime(int off, int bytes) { fetch(off, bytes, parity); }
    // before the workgroup exits: the last prefetch has landed
    SP_DEV void drain() { __syncthreads(); }
    // returns the LDS address of the current chunk; prefetches the next one
    SP_DEV const char* acquire(int next_off, int next_bytes) {
        __syncthreads();               // vmcnt(0) for own DMA + workgroup barrier
        fetch(next_off, next_bytes, parity ^ 1u);
        const char* cur = lds + parity * CHUNK_MAX_BYTES;
        parity ^= 1u;
        return cur;
    }
};

// acc[m] += A(ks, m) * b[ks]   for NKS k-steps and NMB m-blocks of one chunk.
// A fragments are read from LDS PF fragments ahead of the MFMA that consumes them; the
// sched_barrier pins the 1 MFMA : 1 LDS-read source order (left alone, hipcc sinks every
// read next to its MFMA to save registers and exposes the full LDS latency each time).
template <class P, int NMB, int NKS>
SP_DEV void mma_chunk(f32x16 (&acc)[P::G], const typename P::B* b, const char* chunk, int lane) {
    constexpr int N = NKS * NMB;
    constexpr int PF = N < P::PREFETCH ? N : P::PREFETCH;
    typename P::A a[PF];
    const char* base = chunk + lane * P::LANE_BYTES;
#pragma unroll
    for (int i = 0; i < PF; ++i) a[i] = P::lds_frag(base + i * P::FRAG_BYTES);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int ks = i / NMB, m = i % NMB;
        acc[m] = P::mfma(a[i % PF], b[ks], acc[m]);
        if (i + PF < N) a[i % PF] = P::lds_frag(base + (i + PF) * P::FRAG_BYTES);
        __builtin_amdgcn_sched_barrier(0);   // keep source order: MFMA i, then the read for MFMA i+PF
    }
}

// 16-byte store/load of CH activation elements (one chunk of a saved row)
template <class P> SP_DEV void store_chunk(typename P::act_t* row, int c, int h, const typename P::B* v) {
    if constexpr (P::PREC == PREC_BF16) {
        *(bf16x8*)(row + (2 * c + h) * 8) = v[c];
    } else {
        f32x4 t = {v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
        *(f32x4*)(row + (2 * c + h) * 4) = t;
    }
}
template <class P> SP_DEV void load_chunk(const typename P::act_t* row, int c, int h, typename P::B* v) {
    if constexpr (P::PREC == PREC_BF16) {
        v[c] = *(const bf16x8*)(row + (2 * c + h) * 8);
    } else {
        f32x4 t = *(const f32x4*)(row + (2 * c + h) * 4);
        v[4 * c] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
    }
}

// Saved-activation tiles addressed through a raw buffer descriptor (one per saved buffer).
// Tile-major layout (layout.h): for a wave that owns rows 32*T .. 32*T+31,
//   voff = ((T * (cols/CH) + col0/CH) * 32 + n) * 16 + h * 512      (lane n = row&31, half h)
// and k-step chunk c of the vector sits at scalar offset c * 1024: one instruction = 1 KiB.
template <class P> SP_DEV int tile_voff(int64_t tile32, int cols, int col0, int n, int h) {
    return (int)(((tile32 * (cols / P::CH) + col0 / P::CH) * 32 + n) * 16 + h * 512);
}
template <class P> SP_DEV void bstore_chunk(__amdgpu_buffer_rsrc_t r, int voff, int c, const typename P::B* v) {
    u32x4 t;
    if constexpr (P::PREC == PREC_BF16) {
        t = __builtin_bit_cast(u32x4, v[c]);
    } else {
        t[0] = __builtin_bit_cast(unsigned, v[4 * c]); t[1] = __builtin_bit_cast(unsigned, v[4 * c + 1]);
        t[2] = __builtin_bit_cast(unsigned, v[4 * c + 2]); t[3] = __builtin_bit_cast(unsigned, v[4 * c + 3]);
    }
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, c * 1024, 0);
}
// descriptor of saved buffer `b` inside a save / grad area of a pass with `rows` rows
template <class P> SP_DEV __amdgpu_buffer_rsrc_t row_rsrc(const void* area, int64_t rows, int64_t coloff, int cols) {
    const int64_t rp = rows_padded(rows);
    const char* base = (const char*)area + rp * coloff * (int64_t)sizeof(typename P::act_t);
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(rp * cols * (int64_t)sizeof(typename P::act_t)), 0x00020000);
}

// accumulator group initialised with the packed bias of m-blocks [mb0, mb0+NMB); the
// packed bias table lives in LDS (copied once per workgroup), `bias_h` = table + half*64 B
template <class P, int NMB>
SP_DEV void init_acc(f32x16 (&acc)[P::G], const char* bias_h, int layer_float_off, int mb0) {
#pragma unroll
    for (int m = 0; m < NMB; ++m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 t = *(const f32x4*)(bias_h + (layer_float_off + (mb0 + m) * 32) * 4 + k * 16);
            acc[m][4 * k] = t[0]; acc[m][4 * k + 1] = t[1]; acc[m][4 * k + 2] = t[2]; acc[m][4 * k + 3] = t[3];
        }
    }
}
// copy the packed bias table global -> LDS (all threads; caller barriers afterwards)
template <int NTHREADS> SP_DEV void stage_bias(const float* g, char* lds_bias) {
    for (int i = threadIdx.x; i < BIAS_PK_FLOATS; i += NTHREADS) ((float*)lds_bias)[i] = g[i];
}
template <class P, int NMB> SP_DEV void zero_acc(f32x16 (&acc)[P::G]) {
#pragma unroll
    for (int m = 0; m < NMB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
}

// positional-encoding frequencies 2^k * pi exactly as torch computes them in fp32
SP_DEV constexpr float pe_freq(int k) { return (float)(1 << k) * 3.14159274101257324219f; }

}  // namespace sparf
