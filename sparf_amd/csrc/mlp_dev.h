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
    typedef __bf16 act_t;      // element of a saved-activation plane
    typedef __bf16 stage_t;    // element of the per-ray view-encoding workspace and the LDS x0 stash
    static SP_DEV B zero() { B z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.0f; return z; }
    static SP_DEV A lds_frag(const char* p) { return *(const bf16x8*)p; }
    enum { NPART = 1 };
    template <int PART> static SP_DEV f32x16 mfma_part(const A& a, const B& b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static SP_DEV void set(B* v, int q, float x) { v[q >> 3][q & 7] = (__bf16)x; }
    static SP_DEV float get(const B* v, int q) { return (float)v[q >> 3][q & 7]; }
};

template <> struct Policy<PREC_FP32> {
    enum { PREC = PREC_FP32, KJ = 1, CH = 4, FRAG_BYTES = 256, LANE_BYTES = 4, G = group_g(PREC_FP32), NWAVES = nwaves_of(PREC_FP32), PREFETCH = 4 };
    typedef float B;
    typedef float A;
    typedef float act_t;
    typedef float stage_t;
    static SP_DEV B zero() { return 0.0f; }
    static SP_DEV A lds_frag(const char* p) { return *(const float*)p; }
    enum { NPART = 1 };
    template <int PART> static SP_DEV f32x16 mfma_part(const A& a, const B& b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static SP_DEV void set(B* v, int q, float x) { v[q] = x; }
    static SP_DEV float get(const B* v, int q) { return v[q]; }
};

// bf16x3: operands are (head, tail) bf16 pairs, a product is three MFMAs (layout.h PREC_X3)
struct bfpair { bf16x8 hi, lo; };
#ifndef SP_X3_PREFETCH
#define SP_X3_PREFETCH 4     // A-fragment pairs read ahead of their MFMAs (8 VGPRs each)
#endif
template <> struct Policy<PREC_X3> {
    enum { PREC = PREC_X3, KJ = 8, CH = 8, FRAG_BYTES = 2048, LANE_BYTES = 16, G = group_g(PREC_X3), NWAVES = nwaves_of(PREC_X3), PREFETCH = SP_X3_PREFETCH };
    typedef bfpair B;
    typedef bfpair A;
    typedef __bf16 act_t;
    typedef float stage_t;
    static SP_DEV B zero() { B z; for (int i = 0; i < 8; ++i) { z.hi[i] = (__bf16)0.0f; z.lo[i] = (__bf16)0.0f; } return z; }
    static SP_DEV A lds_frag(const char* p) { A a; a.hi = *(const bf16x8*)p; a.lo = *(const bf16x8*)(p + 1024); return a; }
    // the three partial products of one k-step, small terms first
    enum { NPART = 3 };
    template <int PART> static SP_DEV f32x16 mfma_part(const A& a, const B& b, f32x16 c) {
        if constexpr (PART == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
        else if constexpr (PART == 1) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
    }
    static SP_DEV void set(B* v, int q, float x) {
        const __bf16 h = (__bf16)x;
        v[q >> 3].hi[q & 7] = h;
        v[q >> 3].lo[q & 7] = (__bf16)(x - (float)h);
    }
    static SP_DEV float get(const B* v, int q) { return (float)v[q >> 3].hi[q & 7] + (float)v[q >> 3].lo[q & 7]; }
};

// bf16x3 data-gradient chain ("x2"): weights as (head, tail) pairs -- their rounding error would be the
// SAME for every row, i.e. a systematic error of the Jacobian -- but the propagated gradient dY as
// plain bf16: its rounding is unbiased and independent per row, and everything it feeds (weight
// gradients, pose gradients) is a sum over rows / rays.  Two MFMAs per product, bf16 register
// footprint: 8 waves x 256-row tiles like the bf16 mode.  Uses the bf16x3 layout and W^T stream.
#ifndef SP_X3_DGRAD_PARTS
#define SP_X3_DGRAD_PARTS 2      // 1 = experiment: weight heads only (plain bf16 Jacobian)
#endif
// Geometry of the bf16x3 data-gradient kernel: W = 8 waves (two per SIMD inside 256 VGPRs, 256-row workgroup tiles: 12-17 % faster per
// row, mlp_bwd_impl.h bwd_layer_deferred) or W = 4 (one wave per SIMD with the whole register file, 128-row tiles, the forward
// kernels' geometry).  Both are compiled into the library since round 6 (mlp_bwd_x3.hip / mlp_bwd_x3w4.hip) and api.hip picks one
// per launch from the row count: a 512-ray step's coarse pass is 32 768 rows = 128 tiles of 256 rows -- half the chip idle for
// a whole tile time -- but 256 tiles of 128 rows.  SP_X3_DGRAD_WAVES pins one geometry for every launch (A/B builds).
#ifndef SP_X3_DGRAD_WAVES
#define SP_X3_DGRAD_WAVES 0      // 0 = by row count (api.hip x3_dgrad_waves)
#endif
template <int W> struct PolicyX3DgradT {
    enum { PREC = PREC_X3, KJ = 8, CH = 8, FRAG_BYTES = 2048, LANE_BYTES = 16, G = group_g(PREC_X3), NWAVES = W,
#ifdef SP_X3_DGRAD_PREFETCH
           PREFETCH = SP_X3_DGRAD_PREFETCH,
#else
           PREFETCH = (W == 8 ? 3 : 4),
#endif
           NPART = SP_X3_DGRAD_PARTS };
    typedef bf16x8 B;
    typedef bfpair A;
    typedef __bf16 act_t;
    typedef float stage_t;
    static SP_DEV B zero() { B z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.0f; return z; }
    static SP_DEV A lds_frag(const char* p) { A a; a.hi = *(const bf16x8*)p; a.lo = *(const bf16x8*)(p + 1024); return a; }
    template <int PART> static SP_DEV f32x16 mfma_part(const A& a, const B& b, f32x16 c) {
        if constexpr (PART == 0 && NPART == 2) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b, c, 0, 0, 0);
    }
    static SP_DEV void set(B* v, int q, float x) { v[q >> 3][q & 7] = (__bf16)x; }
    static SP_DEV float get(const B* v, int q) { return (float)v[q >> 3][q & 7]; }
};
typedef PolicyX3DgradT<8> PolicyX3Dgrad;          // the 256-row geometry (and the one the 8-bit-area kernels keep)
typedef PolicyX3DgradT<4> PolicyX3DgradW4;

// ------------------------------------------------------------------ wave-time accounting (SP_PROF builds only)
// tools/kernel_bench.py prints where wave 0 of workgroup 0 of the forward kernel spends its cycles
// (s_memtime laps): 0 barrier wait, 1 weight-DMA issue, 2 LDS fragments + MFMA issue, 3 epilogue,
// 4 activation stores, 5 end of tile (sigmoid, output stores, loop), 6 staged tile inputs, 7 encoding (sincos loop), 8 x0 operand
// build, 9 unused.  NOTE slot 0 of a tile's FIRST chunk also holds whatever runs between the last lap and that barrier.
#ifdef SP_PROF
struct Prof {
    unsigned long long last, acc[10];
    SP_DEV void start() { for (int i = 0; i < 10; ++i) acc[i] = 0; last = __builtin_amdgcn_s_memtime(); }
    SP_DEV void lap(int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[k] += now - last; last = now; }
};
#define SP_LAP(prof, k) (prof).lap(k)
#else
struct Prof { SP_DEV void start() {} };
#define SP_LAP(prof, k) ((void)0)
#endif

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
// Fetching two chunks ahead (three buffers, counted wait) in the 4-wave inference kernels,
// whose chunks last only ~0.8 us: also equal (2.08 vs 2.09 ms bf16x3, 6.28 vs 6.29 ms fp32).
// Anti-phase waves (the two waves of a SIMD doing their stores + DMA before / after the chunk's
// MFMAs, three buffers, compile-time counted waits): correct, 1.24-1.27 vs 1.23-1.25 ms -- the
// per-chunk barrier keeps every wave's own VMEM + MFMA chain on the critical path.
// The lane part of the DMA address added by the buffer unit (descriptor ADD_TID_ENABLE, stride 16: `buffer_load_dwordx4 off, ...
// lds` without a VGPR offset): equal (2.23 vs 2.23 ms, profiles/r03h_kernel_ab_dma_tid.log) -- a piece's ~85 issue cycles are
// not its address operand.  What the weight DMA costs the waves that issue it, measured by issuing every piece a second time into
// a dummy LDS area (same results, same data): +4.4 % training / +4.8 % inference forward (profiles/r03h_kernel_ab_dma_twice.log),
// i.e. ~16 cycles per piece -- the 515 pieces a wave issues per tile are not where the time is.
enum { PIPE_LDS_BYTES = 2 * CHUNK_MAX_BYTES };
// timing probe (WRONG RESULTS): with SP_PROBE_NO_DMA the kernels ask the pipe for zero bytes of every next chunk and prime both buffers
// with the first one (mlp_fwd_impl.h / mlp_bwd_impl.h)
#ifdef SP_PROBE_NO_DMA
#define SP_PROBE_NBYTES(x) 0
#else
#define SP_PROBE_NBYTES(x) (x)
#endif

template <int NWAVES, bool SPREAD = false> struct WeightPipe {
    __amdgpu_buffer_rsrc_t rsrc;   // packed stream (global), addressed as raw buffer
    char* lds;                     // 2 * CHUNK_MAX_BYTES
    int wave, lane16;
    unsigned parity;
    Prof prof;

    SP_DEV void init(const char* g, unsigned stream_bytes, char* l) {
        prof.start();
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, stream_bytes, 0x00020000);
        lane16 = (threadIdx.x & 63) * 16;
        lds = l;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    // one 1 KiB piece (index I of this wave) of chunk [OFF, OFF+BYTES) into buffer `buf`.  OFF / BYTES / I are
    // compile-time: as run-time members (round 2) every piece carried a compare + branch + s_and exec (~1150 scalar
    // instructions per 128-row tile of the bf16x3 forward, each an issue slot of the SIMD's only wave)
    template <int OFF, int BYTES, int I> SP_DEV void fetch_piece(unsigned buf) {
        if constexpr (I * NWAVES * 1024 < BYTES) {
            const int o = (I * NWAVES + wave) * 1024;
            if ((I + 1) * NWAVES * 1024 <= BYTES || o < BYTES) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + buf * CHUNK_MAX_BYTES + o), 16,
                                                         lane16, OFF + o, 0, 0);
            }
        }
    }
    enum { PIECES = CHUNK_MAX_BYTES / (NWAVES * 1024), IS_SPREAD = SPREAD };
    SP_DEV void prime(int off, int bytes) { fetch(off, bytes, parity); }
    // before the workgroup exits: the last prefetch has landed
    SP_DEV void drain() { __syncthreads(); }
    // returns the LDS address of the current chunk; prefetches the next one [NOFF, NOFF + NBYTES) (compile-time: whole
    // pieces are issued without a range check -- as run-time arguments every piece carried a compare + branch)
    template <int NOFF, int NBYTES> SP_DEV const char* acquire() {
#ifndef SP_PROBE_NO_BARRIER         // (timing probe, with SP_PROBE_NO_DMA: mlp_bwd_impl.h)
        __syncthreads();               // vmcnt(0) for own DMA + workgroup barrier
#endif
        SP_LAP(prof, 0);
        if constexpr (!SPREAD)         // SPREAD: issued piecewise by SpreadFetch<.., NOFF, NBYTES>
            static_for<PIECES>([&](auto ic) { this->template fetch_piece<NOFF, NBYTES, decltype(ic)::value>(parity ^ 1u); });
        SP_LAP(prof, 1);
        const char* cur = lds + parity * CHUNK_MAX_BYTES;
        parity ^= 1u;
        return cur;
    }
};

// acc[m] += A(ks, m) * b[ks]   for NKS k-steps and NMB m-blocks of one chunk.
// A fragments are read from LDS PF fragments ahead of the MFMA that consumes them; the
// sched_barrier pins the 1 MFMA : 1 LDS-read source order (left alone, hipcc sinks every
// read next to its MFMA to save registers and exposes the full LDS latency each time).
struct NoMid { template <class I, class N> SP_DEV void operator()(I, N) const {} };
// WeightPipe<.., SPREAD = true>: the next chunk's DMA pieces of this wave are issued one at a
// time between the MFMAs of the first three quarters of the current chunk instead of in a burst
// after the barrier.  Wave-time accounting (SP_PROF) shows every vector-memory instruction costs
// ~180 issue cycles wherever it sits, so this only pays where nothing else competes for the
// CU's memory pipe: the inference kernels (bf16 0.74 -> 0.705 ms, bf16x3 2.06 -> 2.01 ms);
// with activation stores in the same interval it is neutral to slightly negative.
#ifndef SP_SPREAD_NUM
#define SP_SPREAD_NUM 3      // the pieces are spread over the first NUM / DEN of the chunk's MFMAs
#define SP_SPREAD_DEN 4
#endif
// Slot balance (bf16x3, NPART = 3): the MFMAs of a chunk run [k-step][part][m-block]; the LDS reads of the next fragments follow
// the LAST part's MFMAs, so those gaps already hold two ds_read_b128 each.  A wave alone on its SIMD hides at most ~5 issue
// slots behind one MFMA (MI355X_MICROARCH.md), so everything else is placed by kind: DMA pieces into last-part gaps (2 reads +
// s_mov m0 + buffer_load = 4-5 slots), deferred-epilogue units (3-4 instructions each, mlp_fwd_impl.h) one per gap into the other
// parts' gaps.  Placed by MFMA index alone (round 3a), a third of the units landed on the read gaps (6-10 slots) while a third
// of the other gaps stayed empty.
#ifndef SP_SLOT_BALANCE
#define SP_SLOT_BALANCE 1
#endif
// index (inside a chunk of [k-step][part][m-block] MFMAs) of the t-th MFMA of the last part
SP_DEV constexpr int last_part_slot(int t, int nmb, int npart) { return (t / nmb) * (npart * nmb) + (npart - 1) * nmb + t % nmb; }
template <class Pipe, int NOFF, int NBYTES, int NMB = 1, int NPART = 1> struct SpreadFetch {
    Pipe& pipe;
    template <class I, class N> SP_DEV void operator()(I, N) const {
        if constexpr (Pipe::IS_SPREAD) {
            constexpr int i = I::value, n = N::value, NP = Pipe::PIECES;
            if constexpr (SP_SLOT_BALANCE && NPART > 1) {
                constexpr int nl = n / NPART, span = SP_SPREAD_NUM * nl / SP_SPREAD_DEN > 0 ? SP_SPREAD_NUM * nl / SP_SPREAD_DEN : 1;     // last-part MFMAs
                static_for<NP>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, t0 = j * span / NP, t = t0 < nl ? t0 : nl - 1;
                    if constexpr (last_part_slot(t, NMB, NPART) == i) pipe.template fetch_piece<NOFF, NBYTES, j>(pipe.parity);
                });
            } else {
                constexpr int span = SP_SPREAD_NUM * n / SP_SPREAD_DEN > 0 ? SP_SPREAD_NUM * n / SP_SPREAD_DEN : 1;
                static_for<NP>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, at0 = j * span / NP, at = at0 < n ? at0 : n - 1;
                    if constexpr (at == i) pipe.template fetch_piece<NOFF, NBYTES, j>(pipe.parity);
                });
            }
        }
    }
};
template <class P, int NMB, int NKS, class Mid = NoMid>
SP_DEV void mma_chunk(f32x16 (&acc)[P::G], const typename P::B* b, const char* chunk, int lane, Mid&& mid = Mid{}) {
    constexpr int N = NKS * NMB;
    constexpr int PF = N < P::PREFETCH ? N : P::PREFETCH;
    static_assert(PF >= NMB || N < P::PREFETCH, "the fragments of one k-step are consumed together");
    typename P::A a[PF];
    const char* base = chunk + lane * P::LANE_BYTES;
#pragma unroll
    for (int i = 0; i < PF; ++i) a[i] = P::lds_frag(base + i * P::FRAG_BYTES);
    // per k-step: every partial product (bf16x3: three) of every m-block, products outermost so
    // that consecutive MFMAs never accumulate into the same registers; a fragment's ring slot is
    // refilled right after its last product
    static_for<NKS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        static_for<P::NPART>([&](auto pc) {
            constexpr int part = decltype(pc)::value;
            static_for<NMB>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = ks * NMB + m;
                acc[m] = P::template mfma_part<part>(a[i % PF], b[ks], acc[m]);
                if constexpr (part == P::NPART - 1 && i + PF < N) a[i % PF] = P::lds_frag(base + (i + PF) * P::FRAG_BYTES);
                mid(std::integral_constant<int, (ks * P::NPART + part) * NMB + m>{}, std::integral_constant<int, N * P::NPART>{});
                __builtin_amdgcn_sched_barrier(0);   // keep source order: MFMA, then the read for the MFMA PF fragments later
            });
        });
    });
}

// ------------------------------------------------------------------ deferred epilogue (one-wave-per-SIMD kernels)
// mlp_fwd_impl.h fwd_layer / mlp_bwd_impl.h bwd_layer_deferred: the accumulators are double-buffered and the epilogue of group g-1
// (STAGES units per register pair, `epi(mb, pair, stage, acc, deferred)`) is issued in the shadows of group g's MFMAs.
SP_DEV constexpr int defer_slot(int p, int np, int ntot) { int at = ((2 * p + 1) * ntot) / (2 * np); return at < ntot ? at : ntot - 1; }
// first pair whose slot is >= gi (pairs are spread evenly over the group's NTOT MFMA slots)
SP_DEV constexpr int defer_first(int gi, int np, int ntot) { int p = 0; while (p < np && defer_slot(p, np, ntot) < gi) ++p; return p; }
// NMB = m-blocks of the CURRENT group: with more than one partial product per k-step the units go, one per gap, behind the MFMAs
// of every part but the last (whose gaps hold the fragment reads and the DMA pieces: mlp_dev.h "Slot balance")
template <class P, class Pipe, class Epi, int NMB_PREV, int MB0_PREV, int BASE, int NTOT, int NOFF, int NBYTES, int NMB, int STAGES> struct DeferredEpi {
    Pipe& pipe;
    Epi& epi;
    const f32x16 (&prev)[P::G];
    static constexpr bool BALANCE = SP_SLOT_BALANCE && P::NPART > 1;
    // eligible MFMA slots of the group before slot gi / in total
    static SP_DEV constexpr int eligible_before(int gi) {
        const int blk = P::NPART * NMB, el = (P::NPART - 1) * NMB, r = gi % blk;
        return BALANCE ? (gi / blk) * el + (r < el ? r : el) : gi;
    }
    template <class I, class N> SP_DEV void operator()(I ic, N nc) const {
        SpreadFetch<Pipe, NOFF, NBYTES, NMB, P::NPART>{pipe}(ic, nc);
        constexpr int gi = BASE + I::value;                     // MFMA index inside the group
        constexpr int NU = NMB_PREV * 8 * STAGES;           // (pair, stage) units of the previous group
        constexpr bool eligible = !BALANCE || (I::value / NMB) % P::NPART != P::NPART - 1;
        if constexpr (eligible) {
            constexpr int e = eligible_before(gi), etot = eligible_before(NTOT);
            constexpr int u0 = defer_first(e, NU, etot), u1 = defer_first(e + 1, NU, etot);      // units due at this slot
            static_for<u1 - u0>([&](auto uc) {
                constexpr int u = u0 + decltype(uc)::value, p = u / STAGES;
                epi(std::integral_constant<int, MB0_PREV + p / 8>{}, std::integral_constant<int, p % 8>{}, std::integral_constant<int, u % STAGES>{},
                    prev[p / 8], std::true_type{});        // (deferred: the accumulator was written at least one MFMA ago)
            });
        }
    }
};


// ------------------------------------------------------------------ per-tile inputs staged through LDS
// What a 32-row tile reads per lane from global memory -- its depth sample, the ray's centre and direction, the ray's
// view-encoding row -- is copied into a per-wave LDS staging area by LDS-DMA long before it is needed (rows: at the top of the
// PREVIOUS tile; view encoding: at the top of its own tile, read before layer 8), so no wave ever waits for a global load with
// its MFMA pipe idle (round 3 profile: the loads sat on s_waitcnt vmcnt(0) at the tile top and at the first barrier of
// layer 8).  The copies are issued from inline asm: invisible to the compiler's vmcnt bookkeeping (extra outstanding
// operations only make its counted waits more conservative -- vector-memory operations return in order) and complete by
// construction, because every weight chunk's barrier waits vmcnt(0) and >= 8 of those lie between issue and use.
//   lds[dst + lane * 4] <- *(dword*)src        lds[dst + lane * 16] <- *(16 bytes*)src     (dst wave-uniform)
SP_DEV unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
SP_DEV void dma_b32(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
SP_DEV void dma_b128(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
enum { STG_ROW_ITEMS = 8, STG_ROW_BYTES = STG_ROW_ITEMS * 256 };       // t, centre xyz, direction xyz, ray index: [item][lane] dwords
// view-encoding row of a ray: 32 staged elements, 16 per lane half = VENC_PIECES 16-byte pieces per lane, staged [piece][lane]
template <class P> struct VencStage {
    enum { EB = (int)sizeof(typename P::stage_t), PIECES = 16 * EB / 16, BYTES = PIECES * 1024, PPC = P::CH * EB / 16 };
    // byte offset of piece q of lane half h inside the ray's row ([16-byte-chunk-of-CH-elements][half] order, load_chunk)
    static SP_DEV int src_off(int q, int h) { return ((2 * (q / PPC) + h) * P::CH) * EB + (q % PPC) * 16; }
};
// the 16 view-encoding elements of this lane half out of the staging area -> B operand chunks (16-byte pieces [piece][lane])
template <class P> SP_DEV void load_chunk(const char* stage, int c, int lane, typename P::B* v) {
    if constexpr (P::PREC == PREC_BF16) {
        v[c] = *(const bf16x8*)(stage + c * 1024 + lane * 16);
    } else if constexpr (P::PREC == PREC_FP32) {
        f32x4 t = *(const f32x4*)(stage + c * 1024 + lane * 16);
        v[4 * c] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
    } else {
        const f32x4 t0 = *(const f32x4*)(stage + (2 * c) * 1024 + lane * 16), t1 = *(const f32x4*)(stage + (2 * c + 1) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { P::set(v, 8 * c + j, t0[j]); P::set(v, 8 * c + 4 + j, t1[j]); }
    }
}

// Save / gradient areas are tile-block-major (layout.h): a wave addresses its 32-row tile through ONE raw
// buffer descriptor, base = area + tile32 * tile bytes.  A chunk block = [row&31][16 B] = 512 B and the two lane
// halves interleave chunk-wise (layout.h pos_of), so 16-byte k-step chunk c of a vector that starts at column
// col0 of buffer b sits at
//     buf_off(b) + (col0 / CH) * 512 + c * 1024   (compile-time scalar offset)   +   h * 512 + n * 16   (lane part)
// The lane part (`lane_voff`) is the same VGPR for every store of the kernel.
// in_area = false: a descriptor of ZERO bytes -- every store through it is dropped and every load returns 0 (raw-buffer range
// check).  Needed by the dgrad kernel when its row range does not start on a workgroup tile: the waves past the end of the range
// then own 32-row tiles beyond the (256-row padded) areas.
template <class P> SP_DEV __amdgpu_buffer_rsrc_t tile_rsrc(const void* area, int64_t tile32, int64_t tile_bytes, bool in_area = true) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)area + (in_area ? tile32 : 0) * tile_bytes), 0, in_area ? (unsigned)tile_bytes : 0u, 0x00020000);
}
SP_DEV int lane_voff(int n, int h) { return n * 16 + h * 512; }
// Cache policy of the activation / gradient saves (buffer-instruction aux bits: 1 = sc0, 2 = nt,
// 16 = sc1).  They are written once and read once, by a later kernel: non-temporal, so that 3.8 GB
// of streaming stores per launch do not displace the 1-2 MB weight stream every CU re-reads from its
// XCD's 4 MB L2.  Same-box A/B (tools/ab_kernels.sh, -DSP_SAVE_AUX=0 is the old default policy),
// bf16: training forward 1.25 -> 1.15 ms, dgrad 0.98 -> 0.88 ms, whole forward pass 1.21 -> 1.03 ms;
// bf16x3: 2.58 -> 2.55, 1.44 -> 1.40 ms.
#ifndef SP_SAVE_AUX
#define SP_SAVE_AUX 2
#endif
// store k-step chunk C of vector v at byte offset BASE (+ C * 1024) of the tile block; PLANE1 = distance of the tail plane
template <class P, int BASE, int C, int PLANE1> SP_DEV void bstore_chunk(__amdgpu_buffer_rsrc_t r, int voff, const typename P::B* v) {
    constexpr int OFF = BASE + C * 1024;
#ifdef SP_PROBE_HALF_SAVES      // timing probe only (wrong results): every other 16-byte store dropped = the store count and bytes of 8-bit saves
    if constexpr (C % 2 == 1) return;
#endif
    if constexpr (sizeof(typename P::B) == 16) {            // one bf16x8 per k-step (bf16, bf16x3 dgrad)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[C]), r, voff, OFF, SP_SAVE_AUX);
    } else if constexpr (P::PREC == PREC_FP32) {
        u32x4 t;
        t[0] = __builtin_bit_cast(unsigned, v[4 * C]); t[1] = __builtin_bit_cast(unsigned, v[4 * C + 1]);
        t[2] = __builtin_bit_cast(unsigned, v[4 * C + 2]); t[3] = __builtin_bit_cast(unsigned, v[4 * C + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, OFF, SP_SAVE_AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[C].hi), r, voff, OFF, SP_SAVE_AUX);
        if constexpr (nplanes_of(PREC_X3) == 2)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[C].lo), r, voff, OFF + PLANE1, SP_SAVE_AUX);
    }
}

// ------------------------------------------------------------------ 8-bit saves (layout.h AREA_Q8)
// A saved vector of a row lives in the wave as bf16x8 k-step chunks, 8 per-lane-half slots each; the row's other half sits in
// lane ^ 32.  q8_amax: max |x| over NCH chunks of this lane and the partner lane, as a float.  On the BIT PATTERNS: non-negative
// bf16 values order like unsigned 16-bit integers (v_pk_max_u16, two elements per instruction); signed vectors drop the sign bits
// first.  hi_plane(v, c) = the bf16x8 image that is saved (bf16x3 forward: the heads).
SP_DEV bf16x8 hi_plane(const bf16x8* v, int c) { return v[c]; }
SP_DEV bf16x8 hi_plane(const bfpair* v, int c) { return v[c].hi; }
template <bool NONNEG, int NCH, class BT> SP_DEV float q8_amax(const BT* v) {
    // (inline asm for the same reason as q8_lo / q8_hi below: as vector code the optimiser rebuilds every packed pair from single
    // conversions of its fp32 sources -- 1 090 extra v_cvt_pk_bf16_f32 and as many v_perm_b32 per tile of the bf16 forward)
    unsigned m = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const u32x4 t = __builtin_bit_cast(u32x4, hi_plane(v, c));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (NONNEG) {
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(t[j]));
            } else {
                unsigned w;
                asm("v_and_b32 %0, 0x7fff7fff, %1" : "=v"(w) : "v"(t[j]));
                asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(w));
            }
        }
    }
    const unsigned top = (m >> 16) > (m & 0xffffu) ? (m >> 16) : (m & 0xffffu);
    const float mine = __builtin_bit_cast(float, top << 16);
    return fmaxf(mine, __shfl_xor(mine, 32));
}
// 127 / amax (the quantiser's factor) and amax / 127 (the step the weight-gradient kernel multiplies back); an all-zero vector
// gets factor 0.  Inf / NaN rows (a diverged network) produce bounded garbage, as they do in the bf16 planes.
SP_DEV float q8_factor(float amax) { return amax > 0.0f ? 127.0f / amax : 0.0f; }
SP_DEV float q8_step(float amax) { return amax * (1.0f / 127.0f); }
// 16 slots (two bf16x8 chunks) -> 16 bytes u = round(x * factor) + 128: x * factor + (1.5 * 2^23 + 128) leaves u in the low byte of
// the float's bit pattern (round-to-nearest-even of the FMA; |x * factor| <= 127 (1 + 2^-23), so no clamp is needed); v_perm_b32
// gathers four low bytes into a dword with three instructions.
// (The two halves of a packed pair are taken apart in inline asm: written as `w << 16` / `w & 0xffff0000` the optimiser looks through
// the v_cvt_pk_bf16_f32 that packed them in the epilogue, converts every element a second time there -- cvt_pk(x, 0) -- and keeps
// those copies alive until the store: 186 spilled registers in the bf16 training forward.)
SP_DEV float q8_lo(unsigned w) { float f; asm("v_lshlrev_b32 %0, 16, %1" : "=v"(f) : "v"(w)); return f; }
SP_DEV float q8_hi(unsigned w) { float f; asm("v_and_b32 %0, 0xffff0000, %1" : "=v"(f) : "v"(w)); return f; }
SP_DEV unsigned q8_quad(unsigned w0, unsigned w1, float f) {
    constexpr float MAGIC = 12583040.0f;      // 1.5 * 2^23 + 128
    const unsigned t0 = __builtin_bit_cast(unsigned, fmaf(q8_lo(w0), f, MAGIC));
    const unsigned t1 = __builtin_bit_cast(unsigned, fmaf(q8_hi(w0), f, MAGIC));
    const unsigned t2 = __builtin_bit_cast(unsigned, fmaf(q8_lo(w1), f, MAGIC));
    const unsigned t3 = __builtin_bit_cast(unsigned, fmaf(q8_hi(w1), f, MAGIC));
    return __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u) | __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);
}
SP_DEV u32x4 q8_pack16(bf16x8 a, bf16x8 b, float f) {
    const u32x4 ta = __builtin_bit_cast(u32x4, a), tb = __builtin_bit_cast(u32x4, b);
    u32x4 o;
    o[0] = q8_quad(ta[0], ta[1], f); o[1] = q8_quad(ta[2], ta[3], f);
    o[2] = q8_quad(tb[0], tb[1], f); o[3] = q8_quad(tb[2], tb[3], f);
    return o;
}
// store slots [16 C16, 16 C16 + 16) of vector v, quantised with factor f, at byte offset BASE (+ C16 * 1024) of the tile block
// (voff = lane_voff(n, h): the same lane part as the bf16 planes' stores -- half h of block C16 is chunk 2 C16 + h)
// A VMEM store of more than 64 bits reads its data registers for a few cycles after it issues; a VALU write to one of them in that
// window needs wait states, which the compiler inserts for its own instructions but NOT in front of the inline-asm unpack of the
// next block (measured: the first dword of a stored block replaced by the next block's first shifted operand, in the second
// half of the workgroup's waves on some tiles).  Hence the explicit s_nop, pinned behind the store.
template <int BASE, int C16, class BT> SP_DEV void q8_store16(__amdgpu_buffer_rsrc_t r, int voff, const BT* v, float f) {
    __builtin_amdgcn_raw_buffer_store_b128(q8_pack16(hi_plane(v, 2 * C16), hi_plane(v, 2 * C16 + 1), f), r, voff, BASE + C16 * 1024, SP_SAVE_AUX);
    asm volatile("s_nop 1");
    __builtin_amdgcn_sched_barrier(0);
}
// the row's step, from both lane halves to the same address (no exec masking)
template <int OFF> SP_DEV void q8_store_step(__amdgpu_buffer_rsrc_t r, int n, float amax) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q8_step(amax)), r, n * 4, OFF, SP_SAVE_AUX);
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
// accumulator group of layer 0 / the skip layer started at  b + w_x p_x + w_y p_y + w_z p_z  (fp32 FMAs): the
// raw-coordinate columns of the encoded point, which the bf16x3 MFMA stream leaves out (streams.h xyz_pk)
template <class P, int NMB>
SP_DEV void init_acc_xyz(f32x16 (&acc)[P::G], const char* bias_h, int layer_float_off, int which, int mb0, float px, float py, float pz) {
#pragma unroll
    for (int m = 0; m < NMB; ++m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = ((mb0 + m) * 32) * 4 + k * 16;
            const f32x4 b = *(const f32x4*)(bias_h + layer_float_off * 4 + e);
            const f32x4 wx = *(const f32x4*)(bias_h + xyz_pk_off(which, 0) * 4 + e);
            const f32x4 wy = *(const f32x4*)(bias_h + xyz_pk_off(which, 1) * 4 + e);
            const f32x4 wz = *(const f32x4*)(bias_h + xyz_pk_off(which, 2) * 4 + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][4 * k + j] = fmaf(wz[j], pz, fmaf(wy[j], py, fmaf(wx[j], px, b[j])));
        }
    }
}
// copy the packed bias table (+ the raw-coordinate columns) global -> LDS (all threads; caller barriers afterwards)
template <int NTHREADS, int NFLOATS> SP_DEV void stage_bias(const float* g, char* lds_bias) {
    for (int i = threadIdx.x; i < NFLOATS; i += NTHREADS) ((float*)lds_bias)[i] = g[i];
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
