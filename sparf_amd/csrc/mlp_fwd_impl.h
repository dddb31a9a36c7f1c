// Fused NeRF MLP forward: sample point -> positional encoding -> 8x256 MLP (+skip) ->
// raw sigma, view branch 283->128->3 -> sigmoid.  One wave owns 32 sample rows and keeps
// their activations in registers across all ten layers (see layout.h); weights stream
// L2 -> LDS (LDS-DMA, double-buffered 32 KiB chunks) and are shared by the workgroup's waves.
// In training each layer stores its INPUT vector (the B operand it holds in registers anyway)
// one slice per accumulator group, in a short burst right after that group's first chunk
// barrier -- the only place where a store's issue slots are free (mlp_dev.h).
//
// Reference semantics: /root/reference/source/models/frequency_nerf.py:149-226
// (compute_raw_density + forward), :47-69/:229-258 (encoding, c2f mask),
// /root/reference/source/utils/camera.py:433-435 (p = c + r*t).
//
// This file is the body of six translation units (mlp_fwd_{bf16,fp32,x3}_{train,infer}.hip: one kernel each, compiled in
// parallel -- as one unit the six kernel instantiations took many minutes).
#pragma once
#include <utility>

#include "kernels.h"
#include "mlp_dev.h"

#if !defined(SP_FWD_PREC) || !defined(SP_FWD_SAVE)
#error "include from mlp_fwd_<precision>_<train|infer>.hip with SP_FWD_PREC / SP_FWD_SAVE defined"
#endif

namespace sparf {

#ifdef SP_PROF
static __device__ unsigned long long g_prof[10];
#endif

// one layer: for each accumulator group, bias init, one chunk per (input segment, k-part),
// epilogue.  save(g, ngroups) runs right after the group's first chunk barrier (the layer's input-vector
// stores; spreading them one at a time over all chunks of the layer measured neutral -- the chip drains the
// saves at ~5 TB/s whatever their spacing, tools/probes/vmem_probe.hip -- and was dropped).
//
// Deferred epilogue (DEFER): turning a group's accumulators into the next layer's B operand -- ReLU, the
// bf16 head / tail split, the sign bits -- is 10-11 VALU instructions per element, ~30 % of a bf16x3 wave's
// time when it runs as its own phase, because the 4-wave kernels have ONE wave per SIMD and nothing else
// to issue meanwhile.  With DEFER the accumulators are double-buffered and group g-1's epilogue is issued
// two elements at a time in the shadows of group g's MFMAs (an MFMA occupies the matrix pipe for 32-64
// cycles, its issue takes ~8: MI355X_MICROARCH.md "single-issue instructions hidden per MFMA gap"); only
// the layer's last group still has an exposed epilogue.  The epilogue of a register pair is cut into EPI_STAGES
// pieces of 3-4 VALU instructions (an MFMA gap hides ~5 single-issue instructions; a whole pair, ~22, issued
// behind one MFMA leaves the matrix pipe idle for two MFMA times -- SQ counters of the first, pair-granular
// version: 24 % of the wave's time issuing VALU with the pipe idle): epi(mb, pair, stage, acc) runs stage
// `stage` of elements 2*pair, 2*pair+1 of m-block mb; the units of a group arrive in order.
enum { EPI_STAGES = 4 };
#ifndef SP_LAZY_ACC_READ
#define SP_LAZY_ACC_READ 1
#endif
// (DeferredEpi, the functor that places the units: mlp_dev.h)
// MFMAs a group issues before chunk (s, kp) / in total
template <class P, int L, int NMB> SP_DEV constexpr int group_mfmas_before(int s_end, int kp_end) {
    int n = 0;
    for (int s = 0; s < layer_nseg(L); ++s)
        for (int kp = 0; kp < fwd_seg_nparts(P::PREC, L, s); ++kp) {
            if (s == s_end && kp == kp_end) return n;
            n += fwd_chunk(P::PREC, fwd_chunk_id(P::PREC, L, 0, s, kp)).nks * NMB * P::NPART;
        }
    return n;
}

struct Xyz { float x, y, z; };      // the sample point (bf16x3: raw-coordinate columns as fp32 FMAs, mlp_dev.h init_acc_xyz)

template <class P, int L, bool DEFER, class Pipe, class Epi, class Save>
SP_DEV void fwd_layer(Pipe& pipe, const char* bias_h, int lane, const typename P::B* in0,
                      const typename P::B* in1, Epi&& epi, Save&& save, const Xyz& pt = Xyz{0.f, 0.f, 0.f}) {
    constexpr int PREC = P::PREC, G = P::G;
    constexpr int NMB_TOT = layer_out_mb(L);
    constexpr int NG = fwd_ngroups(PREC, L);
    f32x16 accs[DEFER ? 2 : 1][G];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int mb0 = g * G;
        constexpr int nmb = (NMB_TOT - mb0) < G ? (NMB_TOT - mb0) : G;
        constexpr int cur_i = DEFER ? (g & 1) : 0, prev_i = DEFER ? ((g & 1) ^ 1) : 0;
        constexpr int nmb_prev = g > 0 ? G : 0;                 // every group but the last is full
        constexpr int ntot = group_mfmas_before<P, L, nmb>(-1, -1);
        f32x16 (&acc)[G] = accs[cur_i];
        if constexpr (xyz_exact(PREC) && (L == 0 || L == 4)) init_acc_xyz<P, nmb>(acc, bias_h, bias_pk_off(L), L == 0 ? 0 : 1, mb0, pt.x, pt.y, pt.z);
        else init_acc<P, nmb>(acc, bias_h, bias_pk_off(L), mb0);
        static_for<layer_nseg(L)>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            static_for<fwd_seg_nparts(PREC, L, s)>([&](auto kc) {
                constexpr int kp = decltype(kc)::value;
                constexpr int id = fwd_chunk_id(PREC, L, g, s, kp);
                constexpr Chunk cur = fwd_chunk(PREC, id);
                constexpr int nxt = (id + 1) % fwd_nchunks(PREC);
                constexpr int noff = (int)fwd_chunk_off(PREC, nxt);
                constexpr int nbytes = SP_PROBE_NBYTES(chunk_bytes(PREC, fwd_chunk(PREC, nxt)));
                const char* ch = pipe.template acquire<noff, nbytes>();
                if constexpr (s == 0 && kp == 0) save(gc, std::integral_constant<int, NG>{});
                SP_LAP(pipe.prof, 4);
                if constexpr (DEFER && g > 0) {
                    constexpr int base = group_mfmas_before<P, L, nmb>(s, kp);
                    mma_chunk<P, nmb, cur.nks>(acc, (s == 0 ? in0 : in1) + cur.ks0, ch, lane,
                                               DeferredEpi<P, Pipe, std::remove_reference_t<Epi>, nmb_prev, mb0 - G, base, ntot, noff, nbytes, nmb, EPI_STAGES>{pipe, epi, accs[prev_i]});
                } else {
                    mma_chunk<P, nmb, cur.nks>(acc, (s == 0 ? in0 : in1) + cur.ks0, ch, lane, SpreadFetch<Pipe, noff, nbytes, nmb, P::NPART>{pipe});
                }
                SP_LAP(pipe.prof, 2);
            });
        });
        if constexpr (!DEFER || g == NG - 1) {
            static_for<nmb * 8 * EPI_STAGES>([&](auto uc) {
                constexpr int u = decltype(uc)::value, p = u / EPI_STAGES;
                epi(std::integral_constant<int, mb0 + p / 8>{}, std::integral_constant<int, p % 8>{}, std::integral_constant<int, u % EPI_STAGES>{},
                    acc[p / 8]);
            });
        }
        SP_LAP(pipe.prof, 3);
    });
}

// SAVEM: 0 inference (nothing saved), 1 training with plane saves, 2 training with 8-bit saves (layout.h AREA_Q8; bf16-operand modes)
template <int PREC, int SAVEM>
__global__ void __launch_bounds__(Policy<PREC>::NWAVES * 64) mlp_fwd_kernel(MlpFwdArgs a) {
    typedef Policy<PREC> P;
    constexpr bool SAVE = SAVEM != 0, Q8 = SAVEM == 2;
    constexpr int AF = area_format(PREC, Q8);          // format of the save area
    static_assert(!Q8 || PREC != PREC_FP32, "8-bit saves: bf16-operand modes only");
    typedef typename P::B B;
    typedef typename P::stage_t stage_t;
    constexpr int KJ = P::KJ, CH = P::CH, NW = P::NWAVES;
    constexpr int NB256 = 128 / KJ, NB128 = 64 / KJ, NBX0 = 32 / KJ, NBV = 16 / KJ;
    constexpr int AUX_FLOATS = xyz_exact(PREC) ? AUX_PK_FLOATS : BIAS_PK_FLOATS;
    // row routing (kernels.h) is compiled into the kernels of the precision far rows run in -- fp32 -- only: the bf16-operand
    // kernels sit at their VGPR limit (tools/kernel_meta.sh) and are never launched routed (api.hip rejects other far precisions)
    constexpr bool ROUTED = PREC == PREC_FP32;

    // LDS image: weight pipe | x0 stash | bias (+ xyz) table | c2f band weights | per-wave staging of the tile inputs
    constexpr int C2F_OFF = PIPE_LDS_BYTES + X0_STASH_BYTES + AUX_FLOATS * 4, STG_OFF = C2F_OFF + 64;
    constexpr int STG_WAVE_BYTES = STG_ROW_BYTES + VencStage<P>::BYTES;
    __shared__ __attribute__((aligned(16))) char lds[STG_OFF + NW * STG_WAVE_BYTES];

    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int64_t BIAS_OFF = packed_bias_off(PREC), FWD_OFF = packed_fwd_off(PREC);
    constexpr unsigned FWD_BYTES = (unsigned)fwd_stream_bytes(PREC);
    constexpr int C0_BYTES = chunk_bytes(PREC, fwd_chunk(PREC, 0));
    stage_bias<NW * 64, AUX_FLOATS>((const float*)(a.packed + BIAS_OFF), lds + PIPE_LDS_BYTES + X0_STASH_BYTES);
    const char* bias_pk = lds + PIPE_LDS_BYTES + X0_STASH_BYTES + h * 64;
    float* c2f = (float*)(lds + C2F_OFF);                 // the ten position-band weights of the pass (c2f_kernel)
    if (threadIdx.x < 10) c2f[threadIdx.x] = a.c2f[threadIdx.x];

    // weight DMA spread between the MFMAs (mlp_dev.h SpreadFetch) in the inference kernels and in the 4-wave
    // training kernels (one wave per SIMD: a burst of 8 pieces after the barrier is time no MFMA is issued;
    // same-box A/B bf16x3 training forward 2.57 -> 2.50 ms); the 8-wave bf16 training kernel keeps the burst
    // (neutral there: its partner wave covers, and its stores compete for the same slots)
    typedef WeightPipe<NW, (!SAVE || NW == 4)> Pipe;
    Pipe pipe;
    pipe.init(a.packed + FWD_OFF, FWD_BYTES, lds);
    pipe.prime(0, C0_BYTES);
#ifdef SP_PROBE_NO_DMA
    pipe.fetch(0, C0_BYTES, 1u);
#endif
    __syncthreads();     // bias table visible to every wave

    const int64_t rows = a.rows;
    const int tile_rows = NW * 32;
    const int64_t ntiles = (rows + tile_rows - 1) / tile_rows;
    const int lvo = lane_voff(n, h);            // lane part of every save address (mlp_dev.h)

    // per-tile inputs through the wave's LDS staging area (mlp_dev.h "per-tile inputs staged through LDS")
    char* stg = lds + STG_OFF + wave * STG_WAVE_BYTES;
    const unsigned stg_a = lds_addr(stg);
    auto stage_rows = [&](int64_t tile_n) {      // depth sample, ray centre / direction and ray index of this lane's row of tile_n
        const int64_t r = (tile_n * NW + wave) * 32 + n;
        const unsigned rc = (unsigned)(r < rows ? r : rows - 1);                 // (rows <= 2^27, sparf_hip.h)
        const unsigned ry = rc / (unsigned)a.nsamp;
        ((unsigned*)stg)[7 * 64 + lane] = ry;
        dma_b32(a.t + (ROUTED ? routed_row(rc, a.nsamp, a.row_stride, a.row_off) : (int64_t)rc), stg_a);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            dma_b32(a.center + 3 * (size_t)ry + j, stg_a + (1 + j) * 256);
            dma_b32(a.dir + 3 * (size_t)ry + j, stg_a + (4 + j) * 256);
        }
    };
    stage_rows(blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the first tile's rows (later ones land a whole tile ahead)

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        SP_LAP(pipe.prof, 5);
        const int64_t tile32 = tile * NW + wave;                 // wave-uniform: this wave's 32-row tile
        const int64_t row = tile32 * 32 + n;
        const bool valid = row < rows;

        // ---- sample point and its encoding (this lane half's 32 of the 64 x0 slots)
        const float* sf = (const float*)stg;
        const float tt = sf[lane];
        const float cx = sf[64 + lane], cy = sf[128 + lane], cz = sf[192 + lane];
        const float dx = sf[256 + lane], dy = sf[320 + lane], dz = sf[384 + lane];
        const unsigned ray = ((const unsigned*)stg)[7 * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staged values are in registers: the area may be refilled
        stage_rows(tile + gridDim.x);                            // next tile of this workgroup (clamped past the end)
        if constexpr (!SAVE) {
            // tile routing by value (kernels.h): is this workgroup tile this launch's to evaluate?  Workgroup-uniform (every wave
            // looks at the same NW values), so all waves skip or none does: the chunk barriers stay matched, and the weight pipe is
            // where a tile leaves it (its last acquire prefetched the first chunk of whatever tile comes next)
            if (a.tile_take != 0) {
                float tmax = -3.0e38f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int64_t r = (tile * NW + w) * 32 + 31;
                    tmax = fmaxf(tmax, a.t[r < rows ? r : rows - 1]);
                }
                if ((tmax > a.tile_thr) != (a.tile_take == 2)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's staged rows have landed (no chunk barrier will wait for them)
                    continue;
                }
            }
        }
        {
            const char* vrow = (const char*)a.venc + (size_t)ray * (32 * sizeof(stage_t));
#pragma unroll
            for (int q = 0; q < VencStage<P>::PIECES; ++q) dma_b128(vrow + VencStage<P>::src_off(q, h), stg_a + STG_ROW_BYTES + q * 1024);
        }
        SP_LAP(pipe.prof, 6);
        const Xyz pt{__fadd_rn(cx, __fmul_rn(dx, tt)), __fadd_rn(cy, __fmul_rn(dy, tt)), __fadd_rn(cz, __fmul_rn(dz, tt))};
        const float px = pt.x, py = pt.y, pz = pt.z;

        // 15 (coord, freq) arguments per lane half, one sincos each, kept in a runtime loop
        // (a single inlined sincosf) and parked in this wave's LDS stash: x0 is needed
        // again by the skip layer and would otherwise pin registers across layers 1-3.
        // half 0: args 0..14 = x:k0..9, y:k0..4 ; half 1: args 15..29 = y:k5..9, z:k0..9
        // Stash image: [slot pair i 0..15][lane 0..63] -- lane-contiguous rows, so every access is a
        // conflict-free ds_*_b32 / b64 (round 2 kept 32 slots per LANE contiguous: a 128-byte lane stride puts
        // all 32 lanes of a group on one bank -- the 12.7 % LDS-conflict share of the forward's PMC profile).
        // bf16: a pair = one dword {bf16 sin, bf16 cos}; fp32 / bf16x3: a pair = two floats (one 8-byte word).
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        char* st = lds + PIPE_LDS_BYTES + wave * (64 * 32 * (int)sizeof(stage_t));
        auto put_pair = [&](int i, float a0, float a1) {
            if constexpr (PREC == PREC_BF16) { const bf16x2_t v = {(__bf16)a0, (__bf16)a1}; ((bf16x2_t*)st)[i * 64 + lane] = v; }
            else { const f32x2 v = {a0, a1}; ((f32x2*)st)[i * 64 + lane] = v; }
        };
        const float pvA = h ? py : px, pvB = h ? pz : py;
        const int split = h ? 5 : 10, kA0 = h ? 5 : 0;
#pragma unroll 1
        for (int i = 0; i < 15; ++i) {
            const bool first = i < split;
            const int k = first ? kA0 + i : i - split;
            const float pv = first ? pvA : pvB;
            const float mk = c2f[k];
            float s, c;
#ifdef SP_PROBE_NO_ENCODING     // timing probe (WRONG RESULTS): what the 15 sincosf per lane half cost the tile -- the upper bound of what a cross-tile
            s = pv * mk; c = mk;    // software pipeline could hide of them (round 6; tools/evidence.sh fwdprobes6)
#else
            sincosf(__fmul_rn(pv, ldexpf(3.14159274101257324219f, k)), &s, &c);
#endif
            put_pair(i, __fmul_rn(s, mk), __fmul_rn(c, mk));
        }
        put_pair(15, h ? pz : px, h ? 0.0f : py);
        SP_LAP(pipe.prof, 7);

        B bx0[NBX0];
        auto load_x0 = [&]() {
            if constexpr (PREC == PREC_BF16) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x4 t;
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = ((const unsigned*)st)[(4 * c + j) * 64 + lane];
                    bx0[c] = __builtin_bit_cast(bf16x8, t);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 v = ((const f32x2*)st)[i * 64 + lane];
                    P::set(bx0, 2 * i, v[0]);
                    P::set(bx0, 2 * i + 1, v[1]);
                }
            }
        };
        load_x0();
        SP_LAP(pipe.prof, 8);

        // this wave's tile block of the save area (layout.h): one descriptor, compile-time offsets inside
        __amdgpu_buffer_rsrc_t srs = pipe.rsrc;
        if constexpr (SAVE) srs = tile_rsrc<P>(a.save, tile32, save_tile_bytes(AF));

        B hA[NB256], hB[NB256];

        // relu epilogue of one register pair (registers 2*pair, 2*pair + 1 of m-block mb) in EPI_STAGES pieces:
        //   0 / 1  ReLU of element 0 / 1 on the BIT PATTERN (v_max_i32: negative floats are negative integers,
        //          -0.0 is INT_MIN; a float max would first canonicalise its input: one more v_max per element),
        //          training: the element's mask bit pushed into the lane's FIFO word (layout.h "ReLU masks":
        //          v_cmp_lt_i32 + v_addc_co_u32, two instructions where compare + select-a-constant + or took three)
        //   2      head: v_cvt_pk_bf16_f32 of the pair (bf16 / bf16x3) or the two fp32 values, into the next layer's B operand
        //   3      bf16x3 tail: bf16(x - float(head)) of the pair; after a layer's last m-block the mask words leave
        //          in ONE 16-byte store per lane
        unsigned mask_bits = 0, e_hi = 0;
        float e_v0 = 0.f, e_v1 = 0.f;
        u32x4 mask_w = {0u, 0u, 0u, 0u};
        auto relu_to = [&](B* out, auto nmbc, auto sbc) {
            return [out, &srs, &mask_w, &mask_bits, &e_hi, &e_v0, &e_v1, lane](auto mbc, auto pairc, auto stagec, const f32x16& acc, auto... deferred) {
                constexpr int mb = decltype(mbc)::value, pr = decltype(pairc)::value, st = decltype(stagec)::value, NMBL = decltype(nmbc)::value;
                constexpr int q0 = 16 * mb + 2 * pr;                 // per-lane-half slot of element 0 (element 1: q0 + 1)
                if constexpr (st == 0 || st == 1) {
                    constexpr int r = 2 * pr + st;
                    float x = acc[r];                                // (bit_cast of a vector-element expression reads element 0)
#if SP_LAZY_ACC_READ
                    // Deferred units fetch their element from the accumulator registers themselves: left to the compiler, all 32
                    // v_accvgpr_read of a finished group are hoisted into the gap at the group boundary (29 instructions next to
                    // the barrier and the first fragment reads: ~240 cycles of idle matrix pipe, 36 times per tile).  Only for
                    // deferred units: the hazard recogniser does not see into asm, and a deferred element was written >= one
                    // whole MFMA (32 cycles; 11 wait states needed) earlier, the exposed epilogue of a layer's last group was not.
                    if constexpr (sizeof...(deferred) > 0 && NW == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(x));     // (volatile: stays in its unit; the 4-wave kernels' accumulators live in AGPRs)
#endif
                    int yi = __builtin_bit_cast(int, x);
                    yi = yi > 0 ? yi : 0;
                    (st == 0 ? e_v0 : e_v1) = __builtin_bit_cast(float, yi);
                    if constexpr (SAVE)
                        asm("v_cmp_lt_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask_bits) : "v"(yi) : "vcc");
                } else if constexpr (st == 2) {
                    if constexpr (PREC == PREC_FP32) {
                        out[q0] = e_v0;
                        out[q0 + 1] = e_v1;
                    } else {
                        const bf16x2_t hp = {(__bf16)e_v0, (__bf16)e_v1};
                        e_hi = __builtin_bit_cast(unsigned, hp);
                        if constexpr (PREC == PREC_BF16) {
                            u32x4 t = __builtin_bit_cast(u32x4, out[q0 >> 3]);
                            t[(q0 & 7) >> 1] = e_hi;
                            out[q0 >> 3] = __builtin_bit_cast(bf16x8, t);
                        } else {
                            u32x4 t = __builtin_bit_cast(u32x4, out[q0 >> 3].hi);
                            t[(q0 & 7) >> 1] = e_hi;
                            out[q0 >> 3].hi = __builtin_bit_cast(bf16x8, t);
                            // x - float(head), element 0.  (One v_dot2c_f32_bf16, x += head . {-1, 0}, instead of shift + subtract
                            // is exact but measured 2 % SLOWER on the whole kernel -- and hipcc folds the {-1, 0} operand into the
                            // inline constant -1.0, which the instruction reads as {0, -1}: tools/probes/dot2_probe.hip.)
                            e_v0 -= __builtin_bit_cast(float, e_hi << 16);
                        }
                    }
                } else {
                    if constexpr (PREC == PREC_X3) {
                        e_v1 -= __builtin_bit_cast(float, e_hi & 0xffff0000u);
                        const bf16x2_t lp = {(__bf16)e_v0, (__bf16)e_v1};
                        u32x4 t = __builtin_bit_cast(u32x4, out[q0 >> 3].lo);
                        t[(q0 & 7) >> 1] = __builtin_bit_cast(unsigned, lp);
                        out[q0 >> 3].lo = __builtin_bit_cast(bf16x8, t);
                    }
                    if constexpr (SAVE && pr == 7 && mb % 2 == 1) {
                        mask_w[mb / 2] = mask_bits;                   // 32 pushes since the last hand-over: the word is complete
#ifndef SP_PROBE_NO_STORES
                        if constexpr (mb == NMBL - 1)
                            __builtin_amdgcn_raw_buffer_store_b128(mask_w, srs, lane * 16, save_mask_tile_off(AF, decltype(sbc)::value), SP_SAVE_AUX);
#endif
                    }
                }
            };
        };
        typedef std::integral_constant<int, 8> MB8;
        typedef std::integral_constant<int, 4> MB4;
        // saver of a layer input: 16-byte chunks [0, NST) of vector v go to columns COL0.. of
        // saved buffer SB; accumulator group g of ng stores its share
        // 8-bit saves: the vector's quantiser factor `qf` is set when group 0 runs (max |x| over the whole vector -- it is complete
        // in registers: it is this layer's B operand -- and the row's step goes out), then every group quantises and stores its share
        // of 16-slot blocks, one 16-byte store each
        float qf_a = 0.0f, qf_b = 0.0f;
        auto saver = [&](auto sbc, auto col0c, auto nstc, const B* v, float& qf) {
            return [&srs, &qf, lvo, n, v](auto gc, auto ngc) {
#ifdef SP_PROBE_NO_STORES      // (timing probe: the mask bits are still computed, nothing is saved)
                return;
#endif
                constexpr int NST = decltype(nstc)::value, g = decltype(gc)::value, ng = decltype(ngc)::value;
                constexpr int sb = decltype(sbc)::value, col0 = decltype(col0c)::value;
                if constexpr (Q8) {
                    constexpr int N16 = NST / 2, b0 = g * N16 / ng, b1 = (g + 1) * N16 / ng;
                    constexpr int part = col0 >= 256 ? 1 : 0;
                    constexpr bool NONNEG = part == 0;                  // ReLU outputs; part 1 = encoded point / view direction
                    constexpr int BASE = save_buf_tile_off(AF, sb) + (col0 / 32) * 1024;
                    if constexpr (g == 0) {
                        const float amax = q8_amax<NONNEG, NST>(v);
                        qf = q8_factor(amax);
                        q8_store_step<save_step_tile_off(sb, part)>(srs, n, amax);
                    }
                    if constexpr (b1 > b0) static_for<b1 - b0>([&](auto cc) { q8_store16<BASE, b0 + decltype(cc)::value>(srs, lvo, v, qf); });
                } else {
                    constexpr int c0 = g * NST / ng, c1 = (g + 1) * NST / ng;
                    constexpr int BASE = save_buf_tile_off(AF, sb) + (col0 / CH) * 512;
                    if constexpr (SAVE && c1 > c0)
                        static_for<c1 - c0>([&](auto cc) { bstore_chunk<P, BASE, c0 + decltype(cc)::value, (int)save_plane_tile_bytes(AF)>(srs, lvo, v); });
                }
            };
        };
        typedef std::integral_constant<int, 32 / CH> NST_X0;
        typedef std::integral_constant<int, 128 / CH> NST_256;
        typedef std::integral_constant<int, 64 / CH> NST_128;
        typedef std::integral_constant<int, 16 / CH> NST_V;
        typedef std::integral_constant<int, 0> C0;
        typedef std::integral_constant<int, 256> C256;
#define SP_SB(b) std::integral_constant<int, b>{}
        // deferred epilogue (fwd_layer) in every kernel.  Rounds 2-4 left the 8-wave bf16 kernels out ("no registers for a second accumulator
        // set, and a partner wave to cover the epilogue") without compiling them: they need 254-256 registers either way and spill LESS with it
        // (5 instead of 8, tools/kernel_meta.sh); bit-identical, training forward 1.061-1.065 -> 1.030-1.045 ms, with 8-bit saves 1.035-1.042 ->
        // 1.005-1.011, inference 0.663 -> 0.654-0.660 (round 5, profiles/r05_kernel_ab_fwd_defer_bf16.log; the data-gradient kernels: mlp_bwd_impl.h)
#ifndef SP_DEFER_EPI
#define SP_DEFER_EPI 1
#endif
        constexpr bool DEFER = SP_DEFER_EPI;

        // (the mask of layer l's OUTPUT lives next to the saved buffer that holds it as the next layer's input)
        { auto e = relu_to(hA, MB8{}, SP_SB(SB_H0)); fwd_layer<P, 0, DEFER, Pipe>(pipe, bias_pk, lane, bx0, bx0, e, saver(SP_SB(SB_XS), C256{}, NST_X0{}, bx0, qf_a), pt); }
        { auto e = relu_to(hB, MB8{}, SP_SB(SB_H1)); fwd_layer<P, 1, DEFER, Pipe>(pipe, bias_pk, lane, hA, hA, e, saver(SP_SB(SB_H0), C0{}, NST_256{}, hA, qf_a)); }
        { auto e = relu_to(hA, MB8{}, SP_SB(SB_H2)); fwd_layer<P, 2, DEFER, Pipe>(pipe, bias_pk, lane, hB, hB, e, saver(SP_SB(SB_H1), C0{}, NST_256{}, hB, qf_a)); }
        { auto e = relu_to(hB, MB8{}, SP_SB(SB_XS)); fwd_layer<P, 3, DEFER, Pipe>(pipe, bias_pk, lane, hA, hA, e, saver(SP_SB(SB_H2), C0{}, NST_256{}, hA, qf_a)); }
        load_x0();
        { auto e = relu_to(hA, MB8{}, SP_SB(SB_H4)); fwd_layer<P, 4, DEFER, Pipe>(pipe, bias_pk, lane, hB, bx0, e, saver(SP_SB(SB_XS), C0{}, NST_256{}, hB, qf_a), pt); }   // h3
        { auto e = relu_to(hB, MB8{}, SP_SB(SB_H5)); fwd_layer<P, 5, DEFER, Pipe>(pipe, bias_pk, lane, hA, hA, e, saver(SP_SB(SB_H4), C0{}, NST_256{}, hA, qf_a)); }
        { auto e = relu_to(hA, MB8{}, SP_SB(SB_H6)); fwd_layer<P, 6, DEFER, Pipe>(pipe, bias_pk, lane, hB, hB, e, saver(SP_SB(SB_H5), C0{}, NST_256{}, hB, qf_a)); }

        // layer 7: C-rows 0..255 -> relu(feat), C-row 256 (block 8, r=0, half 0) -> raw sigma
        float raw_sigma = 0.0f;
        {
            auto relu7 = relu_to(hB, MB8{}, SP_SB(SB_FV));
            auto epi7 = [&](auto mbc, auto pairc, auto stagec, const f32x16& acc, auto... deferred) {
                constexpr int mb = decltype(mbc)::value;
                if constexpr (mb < 8) relu7(mbc, pairc, stagec, acc, deferred...);
                else if constexpr (decltype(pairc)::value == 0 && decltype(stagec)::value == 0) raw_sigma = acc[0];
            };
            fwd_layer<P, 7, DEFER, Pipe>(pipe, bias_pk, lane, hA, hA, epi7, saver(SP_SB(SB_H6), C0{}, NST_256{}, hA, qf_a));
        }
        if (valid && h == 0) a.sigma_raw[ROUTED ? routed_row(row, a.nsamp, a.row_stride, a.row_off) : row] = raw_sigma;     // (kernels.h "row routing")

        // view branch: [feat(256) | view enc(32)] -> 128 -> 3
        B bv[NBV];
        {
#pragma unroll
            for (int c = 0; c < 16 / CH; ++c) load_chunk<P>(stg + STG_ROW_BYTES, c, lane, bv);      // staged at the top of the tile
        }
        B gv[NB128];
        {
            auto s_feat = saver(SP_SB(SB_FV), C0{}, NST_256{}, hB, qf_a);
            auto s_view = saver(SP_SB(SB_FV), C256{}, NST_V{}, bv, qf_b);
            auto e = relu_to(gv, MB4{}, SP_SB(SB_G));
            fwd_layer<P, 8, DEFER, Pipe>(pipe, bias_pk, lane, hB, bv, e, [&](auto gc, auto ngc) { s_feat(gc, ngc); s_view(gc, ngc); });
        }
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        {
            auto epi9 = [&](auto, auto pairc, auto stagec, const f32x16& acc, auto...) {
                if constexpr (decltype(stagec)::value == 0) {
                    if constexpr (decltype(pairc)::value == 0) { z0 = acc[0]; z1 = acc[1]; }
                    else if constexpr (decltype(pairc)::value == 1) z2 = acc[2];
                }
            };
            fwd_layer<P, 9, false, Pipe>(pipe, bias_pk, lane, gv, gv, epi9, saver(SP_SB(SB_G), C0{}, NST_128{}, gv, qf_a));
        }
#undef SP_SB
        SP_LAP(pipe.prof, 9);
#ifdef SP_PROBE_NO_TILE_END     // timing probe (WRONG RESULTS): the tile's last phase -- sigmoid + colour stores -- reduced to one store (keeps z0..z2 live)
        if (valid && h == 0 && z0 + z1 + z2 == 12345.678f) a.rgb[row * 3] = z0;
#else
        if (valid && h == 0) {
            float* o = a.rgb + (ROUTED ? routed_row(row, a.nsamp, a.row_stride, a.row_off) : row) * 3;
            o[0] = 1.0f / (1.0f + expf(-z0));
            o[1] = 1.0f / (1.0f + expf(-z1));
            o[2] = 1.0f / (1.0f + expf(-z2));
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pipe.drain();      // the last prefetches land before the workgroup gives up its LDS
#ifdef SP_PROF
    SP_LAP(pipe.prof, 5);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 10; ++i) g_prof[i] = pipe.prof.acc[i];
#endif
}

int SP_FWD_LAUNCHER(const MlpFwdArgs& a, int grid, hipStream_t stream) {
    if (a.rows <= 0) return 0;
    hipLaunchKernelGGL((mlp_fwd_kernel<SP_FWD_PREC, (int)SP_FWD_SAVE>), dim3(grid), dim3(Policy<SP_FWD_PREC>::NWAVES * 64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf

#if defined(SP_PROF) && SP_FWD_PROF_EXPORT
extern "C" int sparf_debug_prof(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sparf::g_prof), 10 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
