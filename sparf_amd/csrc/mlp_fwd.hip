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
#include <utility>

#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

#ifdef SP_PROF
__device__ unsigned long long g_prof[8];
#endif

// one layer: for each accumulator group, bias init, one chunk per (input segment, k-part),
// epilogue.  save(g, ngroups) runs right after the group's first chunk barrier.
//
// SP_SAVE_SPREAD: the layer's NST input-vector stores are instead spread one at a time over ALL chunks
// of the layer, each in the middle of a run of MFMAs (SpreadStore below).  tools/probes/vmem_probe.hip:
// the chip drains these saves at ~5 TB/s = ~100 cycles per 1 KiB store instruction per CU, and a wave that
// issues a store while that path is backed up simply blocks.  A burst of 4 per wave at every group start
// (all CUs in step) is such a back-up; one store every ~24 MFMAs (bf16x3: 16 KiB per wave and layer over
// 12 k MFMA cycles) is a third of the drain rate and finds the path empty.
template <class Pipe, class StoreOne, int NST, int CI, int NCH> struct SpreadStore {
    Pipe& pipe;
    const StoreOne& store_one;
    template <class I, class N> SP_DEV void operator()(I ic, N nc) const {
        SpreadFetch<Pipe>{pipe}(ic, nc);
        constexpr int i = I::value, n = N::value;
        constexpr int j0 = CI * NST / NCH, j1 = (CI + 1) * NST / NCH, cnt = j1 - j0;
        static_for<(cnt > 0 ? cnt : 0)>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int at0 = (2 * j + 1) * n / (2 * cnt), at = at0 < n ? at0 : n - 1;
            if constexpr (at == i) store_one(std::integral_constant<int, j0 + j>{});
        });
    }
};

template <class P, int L, class Pipe, class Epi, class Save, class StoreOne = int, int NST = 0>
SP_DEV void fwd_layer(Pipe& pipe, const char* bias_h, int lane, const typename P::B* in0,
                      const typename P::B* in1, Epi&& epi, Save&& save, const StoreOne& store_one = 0, std::integral_constant<int, NST> = {}) {
    constexpr int PREC = P::PREC, G = P::G;
    constexpr int NMB_TOT = layer_out_mb(L);
    constexpr int NG = fwd_ngroups(PREC, L);
    constexpr int CPG = fwd_chunks_per_group(PREC, L), NCH = NG * CPG;
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int mb0 = g * G;
        constexpr int nmb = (NMB_TOT - mb0) < G ? (NMB_TOT - mb0) : G;
        f32x16 acc[G];
        init_acc<P, nmb>(acc, bias_h, bias_pk_off(L), mb0);
        static_for<layer_nseg(L)>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            static_for<fwd_seg_nparts(PREC, L, s)>([&](auto kc) {
                constexpr int kp = decltype(kc)::value;
                constexpr int id = fwd_chunk_id(PREC, L, g, s, kp);
                constexpr Chunk cur = fwd_chunk(PREC, id);
                constexpr int nxt = (id + 1) % fwd_nchunks(PREC);
                constexpr int noff = (int)fwd_chunk_off(PREC, nxt);
                constexpr int nbytes = chunk_bytes(PREC, fwd_chunk(PREC, nxt));
                constexpr int ci = id - fwd_chunk_id(PREC, L, 0, 0, 0);      // chunk index inside the layer
                const char* ch = pipe.acquire(noff, nbytes);
                if constexpr (NST == 0) {
                    if constexpr (s == 0 && kp == 0) save(gc, std::integral_constant<int, NG>{});
                    SP_LAP(pipe.prof, 4);
                    mma_chunk<P, nmb, cur.nks>(acc, (s == 0 ? in0 : in1) + cur.ks0, ch, lane, SpreadFetch<Pipe>{pipe});
                } else {
                    mma_chunk<P, nmb, cur.nks>(acc, (s == 0 ? in0 : in1) + cur.ks0, ch, lane, SpreadStore<Pipe, StoreOne, NST, ci, NCH>{pipe, store_one});
                }
                SP_LAP(pipe.prof, 2);
            });
        });
        static_for<nmb>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            epi(std::integral_constant<int, mb0 + m>{}, acc[m]);
        });
        SP_LAP(pipe.prof, 3);
    });
}

template <int PREC, bool SAVE>
__global__ void __launch_bounds__(Policy<PREC>::NWAVES * 64) mlp_fwd_kernel(MlpFwdArgs a) {
    typedef Policy<PREC> P;
    typedef typename P::B B;
    typedef typename P::stage_t stage_t;
    constexpr int KJ = P::KJ, CH = P::CH, NW = P::NWAVES;
    constexpr int NB256 = 128 / KJ, NB128 = 64 / KJ, NBX0 = 32 / KJ, NBV = 16 / KJ;

    __shared__ __attribute__((aligned(16))) char lds[PIPE_LDS_BYTES + X0_STASH_BYTES + BIAS_PK_FLOATS * 4];

    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int64_t BIAS_OFF = packed_bias_off(PREC), FWD_OFF = packed_fwd_off(PREC);
    constexpr unsigned FWD_BYTES = (unsigned)fwd_stream_bytes(PREC);
    constexpr int C0_BYTES = chunk_bytes(PREC, fwd_chunk(PREC, 0));
    stage_bias<NW * 64>((const float*)(a.packed + BIAS_OFF), lds + PIPE_LDS_BYTES + X0_STASH_BYTES);
    const char* bias_pk = lds + PIPE_LDS_BYTES + X0_STASH_BYTES + h * 64;
    const float* c2f = a.c2f;

    // weight DMA spread between the MFMAs (mlp_dev.h SpreadFetch) in the inference kernels and in the 4-wave
    // training kernels (one wave per SIMD: a burst of 8 pieces after the barrier is time no MFMA is issued;
    // same-box A/B bf16x3 training forward 2.57 -> 2.50 ms); the 8-wave bf16 training kernel keeps the burst
    // (neutral there: its partner wave covers, and its stores compete for the same slots)
    typedef WeightPipe<NW, (!SAVE || NW == 4)> Pipe;
    Pipe pipe;
    pipe.init(a.packed + FWD_OFF, FWD_BYTES, lds);
    pipe.prime(0, C0_BYTES);
    __syncthreads();     // bias table visible to every wave
#ifdef SP_PRIO_HALF
    // the second-dispatched half of an 8-wave workgroup loses every VALU / issue arbitration against its
    // SIMD partner (MI355X_MICROARCH.md "Two waves per SIMD" item 4): one static s_setprio for that half
    if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

    const int64_t rows = a.rows;
    const int tile_rows = NW * 32;
    const int64_t ntiles = (rows + tile_rows - 1) / tile_rows;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        SP_LAP(pipe.prof, 5);
        const int64_t tile32 = tile * NW + wave;                 // wave-uniform: this wave's 32-row tile
        const int64_t row = tile32 * 32 + n;
        const bool valid = row < rows;
        const int64_t rowc = valid ? row : rows - 1;
        const int64_t ray = rowc / a.nsamp;

        // ---- sample point and its encoding (this lane half's 32 of the 64 x0 slots)
        const float tt = a.t[rowc];
        const float cx = a.center[ray * 3 + 0], cy = a.center[ray * 3 + 1], cz = a.center[ray * 3 + 2];
        const float dx = a.dir[ray * 3 + 0], dy = a.dir[ray * 3 + 1], dz = a.dir[ray * 3 + 2];
        const float px = __fadd_rn(cx, __fmul_rn(dx, tt));
        const float py = __fadd_rn(cy, __fmul_rn(dy, tt));
        const float pz = __fadd_rn(cz, __fmul_rn(dz, tt));

        // 15 (coord, freq) arguments per lane half, one sincos each, kept in a runtime loop
        // (a single inlined sincosf) and parked in this wave's LDS stash: x0 is needed
        // again by the skip layer and would otherwise pin registers across layers 1-3.
        // half 0: args 0..14 = x:k0..9, y:k0..4 ; half 1: args 15..29 = y:k5..9, z:k0..9
        stage_t* st = (stage_t*)(lds + PIPE_LDS_BYTES) + (wave * 64 + lane) * 32;
#pragma unroll 1
        for (int i = 0; i < 15; ++i) {
            const int arg = 15 * h + i;
            const int coord = arg >= 20 ? 2 : arg >= 10 ? 1 : 0;
            const int k = arg - 10 * coord;
            const float pv = coord == 0 ? px : coord == 1 ? py : pz;
            const float mk = c2f[k];
            float s, c;
            sincosf(__fmul_rn(pv, ldexpf(3.14159274101257324219f, k)), &s, &c);
            st[2 * i] = (stage_t)__fmul_rn(s, mk);
            st[2 * i + 1] = (stage_t)__fmul_rn(c, mk);
        }
        st[30] = (stage_t)(h ? pz : px);
        st[31] = (stage_t)(h ? 0.0f : py);

        B bx0[NBX0];
        auto load_x0 = [&]() {
#pragma unroll
            for (int q = 0; q < 32; q += CH) {
                if constexpr (PREC == PREC_BF16) {
                    bx0[q / 8] = *(const bf16x8*)(st + q);
                } else if constexpr (PREC == PREC_FP32) {
                    f32x4 v = *(const f32x4*)(st + q);
                    bx0[q] = v[0]; bx0[q + 1] = v[1]; bx0[q + 2] = v[2]; bx0[q + 3] = v[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) P::set(bx0, q + j, st[q + j]);
                }
            }
        };
        load_x0();

        // saved-activation tiles: buffers are padded to whole workgroup tiles (layout.h rows_padded),
        // so every wave stores its 32-row tile unconditionally (rows past the end hold the clamped last row)

        B hA[NB256], hB[NB256];

        // relu epilogue; in training also records the sign pattern of the m-block (bit r of
        // 16 bits per lane); two consecutive m-blocks share one 32-bit word and one store
        unsigned* mask_base = nullptr;
        unsigned mask_lo = 0;
        auto relu_to = [&](B* out) {
            return [out, &mask_base, &mask_lo, lane](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    P::set(out, 16 * mb + r, fmaxf(acc[r], 0.0f));
                    if constexpr (SAVE) bits |= (acc[r] > 0.0f ? 1u : 0u) << r;
                }
                if constexpr (SAVE) {
                    if constexpr (mb % 2 == 0) mask_lo = bits;
                    else {
#if SP_SAVE_AUX == 2
                        __builtin_nontemporal_store(mask_lo | (bits << 16), mask_base + (mb / 2) * 64 + lane);
#else
                        mask_base[(mb / 2) * 64 + lane] = mask_lo | (bits << 16);
#endif
                    }
                }
            };
        };
        auto mask_of = [&](int sb) {
            if constexpr (SAVE)
                mask_base = (unsigned*)((char*)a.save + mask_area_off(rows, save_abytes_of(PREC)) + mask_buf_off(rows, sb) +
                                              tile32 * MASK_TILE_BYTES);
        };
        // saver of a layer input: 16-byte chunks [0, NST) of vector v go to columns col0.. of
        // saved buffer sb (row_cols wide); accumulator group g of ng stores its share
        auto saver = [&](int sb, int row_cols, int col0, auto nstc, const B* v) {
            const int vo = tile_voff<P>(tile32, row_cols, col0, n, h);
            const RowRsrc<P> r = row_rsrc<P>(a.save, rows, save_coloff(sb), row_cols, SAVE_COLS);
            return [vo, r, v](auto gc, auto ngc) {
                constexpr int NST = decltype(nstc)::value, g = decltype(gc)::value, ng = decltype(ngc)::value;
                constexpr int c0 = g * NST / ng, c1 = (g + 1) * NST / ng;
                if constexpr (SAVE && c1 > c0) {
#pragma unroll
                    for (int c = c0; c < c1; ++c) bstore_chunk<P>(r, vo, c, v);
                }
            };
        };
        typedef std::integral_constant<int, 32 / CH> NST_X0;
        typedef std::integral_constant<int, 128 / CH> NST_256;
        typedef std::integral_constant<int, 64 / CH> NST_128;
        typedef std::integral_constant<int, 16 / CH> NST_V;
#ifdef SP_SAVE_SPREAD
        // one store at a time (fwd_layer / SpreadStore): 16-byte chunk j of vector v -> columns col0.. of saved buffer sb
        auto one = [&](int sb, int row_cols, int col0, const B* v) {
            const int vo = tile_voff<P>(tile32, row_cols, col0, n, h);
            const RowRsrc<P> r = row_rsrc<P>(a.save, rows, save_coloff(sb), row_cols, SAVE_COLS);
            return [vo, r, v](auto jc) { bstore_chunk<P>(r, vo, decltype(jc)::value, v); };
        };
        auto none = [](auto, auto) {};
#define SP_NST(N) std::integral_constant<int, SAVE ? (N) : 0>{}
#define SP_LAYER(L, IN0, IN1, EPI, SAVER, ONE, N) fwd_layer<P, L, Pipe>(pipe, bias_pk, lane, IN0, IN1, EPI, none, ONE, SP_NST(N))
#else
#define SP_LAYER(L, IN0, IN1, EPI, SAVER, ONE, N) fwd_layer<P, L, Pipe>(pipe, bias_pk, lane, IN0, IN1, EPI, SAVER)
#endif

        mask_of(SB_H0);
        SP_LAYER(0, bx0, bx0, relu_to(hA), saver(SB_XS, 320, 256, NST_X0{}, bx0), one(SB_XS, 320, 256, bx0), NST_X0::value);
        mask_of(SB_H1);
        SP_LAYER(1, hA, hA, relu_to(hB), saver(SB_H0, 256, 0, NST_256{}, hA), one(SB_H0, 256, 0, hA), NST_256::value);
        mask_of(SB_H2);
        SP_LAYER(2, hB, hB, relu_to(hA), saver(SB_H1, 256, 0, NST_256{}, hB), one(SB_H1, 256, 0, hB), NST_256::value);
        mask_of(SB_XS);
        SP_LAYER(3, hA, hA, relu_to(hB), saver(SB_H2, 256, 0, NST_256{}, hA), one(SB_H2, 256, 0, hA), NST_256::value);
        load_x0();
        mask_of(SB_H4);
        SP_LAYER(4, hB, bx0, relu_to(hA), saver(SB_XS, 320, 0, NST_256{}, hB), one(SB_XS, 320, 0, hB), NST_256::value);   // h3
        mask_of(SB_H5);
        SP_LAYER(5, hA, hA, relu_to(hB), saver(SB_H4, 256, 0, NST_256{}, hA), one(SB_H4, 256, 0, hA), NST_256::value);
        mask_of(SB_H6);
        SP_LAYER(6, hB, hB, relu_to(hA), saver(SB_H5, 256, 0, NST_256{}, hB), one(SB_H5, 256, 0, hB), NST_256::value);

        // layer 7: C-rows 0..255 -> relu(feat), C-row 256 (block 8, r=0, half 0) -> raw sigma
        float raw_sigma = 0.0f;
        mask_of(SB_FV);
        auto epi7 = [&](auto mbc, const f32x16& acc) {
            constexpr int mb = decltype(mbc)::value;
            if constexpr (mb < 8) {
                relu_to(hB)(mbc, acc);
            } else {
                raw_sigma = acc[0];
            }
        };
        SP_LAYER(7, hA, hA, epi7, saver(SB_H6, 256, 0, NST_256{}, hA), one(SB_H6, 256, 0, hA), NST_256::value);
        if (valid && h == 0) a.sigma_raw[row] = raw_sigma;

        // view branch: [feat(256) | view enc(32)] -> 128 -> 3
        B bv[NBV];
        {
            const stage_t* vr = (const stage_t*)a.venc + ray * 32;
#pragma unroll
            for (int c = 0; c < 16 / CH; ++c) load_chunk<P>(vr, c, h, bv);
        }
        B gv[NB128];
        mask_of(SB_G);
        {
#ifdef SP_SAVE_SPREAD
            auto o_feat = one(SB_FV, 288, 0, hB);
            auto o_view = one(SB_FV, 288, 256, bv);
            auto o_both = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j < NST_256::value) o_feat(jc);
                else o_view(std::integral_constant<int, j - NST_256::value>{});
            };
            fwd_layer<P, 8, Pipe>(pipe, bias_pk, lane, hB, bv, relu_to(gv), none, o_both, SP_NST(NST_256::value + NST_V::value));
#else
            auto s_feat = saver(SB_FV, 288, 0, NST_256{}, hB);
            auto s_view = saver(SB_FV, 288, 256, NST_V{}, bv);
            fwd_layer<P, 8, Pipe>(pipe, bias_pk, lane, hB, bv, relu_to(gv), [&](auto gc, auto ngc) { s_feat(gc, ngc); s_view(gc, ngc); });
#endif
        }
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        auto epi9 = [&](auto, const f32x16& acc) {
            z0 = acc[0]; z1 = acc[1]; z2 = acc[2];
        };
        SP_LAYER(9, gv, gv, epi9, saver(SB_G, 128, 0, NST_128{}, gv), one(SB_G, 128, 0, gv), NST_128::value);
#undef SP_LAYER
        if (valid && h == 0) {
            float* o = a.rgb + row * 3;
            o[0] = 1.0f / (1.0f + expf(-z0));
            o[1] = 1.0f / (1.0f + expf(-z1));
            o[2] = 1.0f / (1.0f + expf(-z2));
        }
    }
    pipe.drain();      // the last prefetches land before the workgroup gives up its LDS
#ifdef SP_PROF
    SP_LAP(pipe.prof, 5);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) g_prof[i] = pipe.prof.acc[i];
#endif
}

int launch_mlp_fwd(int prec, bool save, const MlpFwdArgs& a, int grid, hipStream_t stream) {
    if (a.rows <= 0) return 0;
#define SP_LAUNCH(PR, SV) \
    hipLaunchKernelGGL((mlp_fwd_kernel<PR, SV>), dim3(grid), dim3(Policy<PR>::NWAVES * 64), 0, stream, a)
    if (prec == PREC_BF16) { if (save) SP_LAUNCH(PREC_BF16, true); else SP_LAUNCH(PREC_BF16, false); }
    else if (prec == PREC_FP32) { if (save) SP_LAUNCH(PREC_FP32, true); else SP_LAUNCH(PREC_FP32, false); }
    else if (prec == PREC_X3) { if (save) SP_LAUNCH(PREC_X3, true); else SP_LAUNCH(PREC_X3, false); }
    else return 1;
#undef SP_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf

#ifdef SP_PROF
extern "C" int sparf_debug_prof(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sparf::g_prof), 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
