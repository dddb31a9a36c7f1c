// Fused NeRF MLP backward, data-gradient half ("dgrad"): from d raw_sigma and d z (colour
// pre-sigmoid) of every sample back through the ten layers, applying the ReLU masks of
// the activations saved by the forward kernel.  Same register-resident scheme as the
// forward (layout.h): dY_l^T is the MFMA B operand, W_l^T fragments stream through LDS.
// Every layer's pre-activation gradient dY_l is written to HBM for the weight-gradient
// kernel (wgrad.hip); with POSE the gradient w.r.t. the sample point (through the
// positional encoding) and the encoded view direction are produced as well.
//
// Math: SURVEY.md Appendix A "Backward" (autograd of
// /root/reference/source/models/frequency_nerf.py:149-226).
#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

// acc += W_l^T[m-group g of segment S] * dY, over all K parts.
// `pre(g)` runs right after the group's first chunk barrier (group 0 loads the layer's ReLU
// mask words there), then `store(g, ngroups)` issues this group's slice of the layer's dY
// stores: a short burst while the wave waits for its first LDS fragments (mlp_dev.h).
template <class P, int L, int S, int GI, int NG, bool POSE, int NMB, class Pre, class Store>
SP_DEV void bwd_group(WeightPipe<P::NWAVES>& pipe, int lane, const typename P::B* dy, f32x16 (&acc)[P::G], Pre&& pre, Store&& store) {
    constexpr int PREC = P::PREC;
    static_for<bwd_nparts(PREC, L)>([&](auto pc) {
        constexpr int part = decltype(pc)::value;
        constexpr int id = bwd_chunk_id(PREC, L, S, GI, part);
        constexpr Chunk cur = bwd_chunk(PREC, id);
        constexpr int nxt = bwd_next_id(PREC, id, POSE);
        constexpr int noff = (int)bwd_chunk_off(PREC, nxt);
        constexpr int nbytes = chunk_bytes(PREC, bwd_chunk(PREC, nxt));
        const char* ch = pipe.acquire(noff, nbytes);
        if constexpr (part == 0) {
            pre(std::integral_constant<int, GI>{});        // accumulators not live yet
            store(std::integral_constant<int, GI>{}, std::integral_constant<int, NG>{});
            zero_acc<P, NMB>(acc);
        }
        mma_chunk<P, cur.nmb, cur.nks>(acc, dy + cur.ks0, ch, lane);
    });
}

template <int PREC, bool POSE, class P = Policy<PREC>>
__global__ void __launch_bounds__(P::NWAVES * 64) mlp_bwd_kernel(MlpBwdArgs a) {
    typedef typename P::B B;
    constexpr int KJ = P::KJ, CH = P::CH, NW = P::NWAVES, G = P::G;
    constexpr int NB256 = 128 / KJ, NB128 = 64 / KJ;

    __shared__ __attribute__((aligned(16))) char lds[PIPE_LDS_BYTES + (POSE ? NW * 64 * 32 * 4 : 16)];

    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int64_t BWD_OFF = packed_bwd_off(PREC);
    constexpr unsigned BWD_BYTES = (unsigned)bwd_stream_bytes(PREC);
    constexpr int C0_BYTES = chunk_bytes(PREC, bwd_chunk(PREC, 0));
    const float* c2f = a.c2f;

    WeightPipe<NW> pipe;
    pipe.init(a.packed + BWD_OFF, BWD_BYTES, lds);
    pipe.prime(0, C0_BYTES);

    const int64_t rows = a.rows;
    const int tile_rows = NW * 32;
    const int64_t ntiles = (rows + tile_rows - 1) / tile_rows;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // tile-major saved buffers (layout.h): this wave's 32 rows form tile `tile32` (wave-uniform)
        const int64_t tile32 = tile * NW + wave;
        const int64_t row = tile32 * 32 + n;
        const bool valid = row < rows;
        const int64_t rowc = valid ? row : rows - 1;
        const int64_t tile_c = tile32;             // buffers are padded to whole workgroup tiles (layout.h rows_padded)
        // ReLU mask words of this lane for saved buffer sb (layout.h "ReLU masks")
        auto load_mask = [&](int sb) {
            return (const unsigned*)((const char*)a.save + mask_area_off(rows, save_abytes_of(PREC)) + mask_buf_off(rows, sb) +
                                           tile_c * MASK_TILE_BYTES) + lane * 4;
        };
        // 16-byte chunks [0, NST) of gradient vector v -> columns col0.. of grad buffer gb;
        // accumulator group g of ng stores its share
        auto store_slice = [&](int gb, int cols, int col0, auto nstc, const B* v) {
            const int vo = tile_voff<P>(tile_c, cols, col0, n, h);
            const RowRsrc<P> r = row_rsrc<P>(a.grad, rows, grad_coloff(gb), cols, GRAD_COLS);
            return [vo, r, v](auto gc, auto ngc) {
                constexpr int NST = decltype(nstc)::value, g = decltype(gc)::value, ng = decltype(ngc)::value;
                constexpr int c0 = g * NST / ng, c1 = (g + 1) * NST / ng;
                if constexpr (c1 > c0) {
#pragma unroll
                    for (int c = c0; c < c1; ++c) bstore_chunk<P>(r, vo, c, v);
                }
            };
        };
        typedef std::integral_constant<int, 128 / CH> NST_256;
        typedef std::integral_constant<int, 64 / CH> NST_128;
        typedef std::integral_constant<int, 16 / CH> NST_16;
        // group 0 first loads the layer's ReLU mask words (layout.h: four 32-bit words per lane, one per
        // m-block pair, written by the forward kernel as one 16-byte store) -- BEFORE the layer's stores, so
        // that waiting for them later does not wait for these stores (vmcnt retires in issue order)
        auto masks_of = [&](const unsigned* mw, unsigned* mk, auto nmc) {
            return [mw, mk](auto gc) {
                if constexpr (decltype(gc)::value == 0) {
                    const u32x4 w = __builtin_nontemporal_load((const u32x4*)mw);
#pragma unroll
                    for (int p = 0; p < decltype(nmc)::value / 2; ++p) mk[p] = w[p];
                }
            };
        };
        // epilogue: dy_prev[q] = acc * [saved activation > 0]
        auto masked_to = [&](const unsigned* mk, B* out) {
            return [mk, out](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
                const unsigned bits = mk[mb / 2] >> (16 * (mb % 2));
#pragma unroll
                for (int r = 0; r < 16; ++r) P::set(out, 16 * mb + r, (bits >> r) & 1u ? acc[r] : 0.0f);
            };
        };
        typedef std::integral_constant<int, 8> NM8;
        typedef std::integral_constant<int, 4> NM4;
        // run all m-groups of segment S of layer L with epilogue epi(mb, acc)
#define SP_BWD_LAYER(L, S, DY, EPI, PRE, STORE)                                                 \
        static_for<bwd_seg_ngroups(PREC, L, S)>([&](auto gc) {                               \
            constexpr int g = decltype(gc)::value;                                           \
            constexpr int tot = vk_width(layer_seg_kind(L, S)) / 32;                         \
            constexpr int nmb = (tot - g * G) < G ? (tot - g * G) : G;                       \
            f32x16 acc[G];                                                                   \
            bwd_group<P, L, S, g, bwd_seg_ngroups(PREC, L, S), POSE, nmb>(pipe, lane, DY, acc, PRE, STORE); \
            static_for<nmb>([&](auto mc) {                                                   \
                constexpr int m = decltype(mc)::value;                                       \
                EPI(std::integral_constant<int, g * G + m>{}, acc[m]);                       \
            });                                                                              \
        })

        // ---- inputs: d z (3, on half 0) and d raw_sigma
        float dz0 = 0.f, dz1 = 0.f, dz2 = 0.f, dsig = 0.f;
        if (valid && h == 0) {
            dz0 = a.d_z[row * 3]; dz1 = a.d_z[row * 3 + 1]; dz2 = a.d_z[row * 3 + 2];
            dsig = a.d_sigma_raw[row];
        }
        B bdz[16 / KJ];
#pragma unroll
        for (int q = 0; q < 16; ++q) P::set(bdz, q, q == 0 ? dz0 : q == 1 ? dz1 : q == 2 ? dz2 : 0.0f);
        auto no_pre = [](auto) {};
        auto no_store = [](auto, auto) {};

        // ---- rgb layer 1 (128 -> 3), transposed: dg = R1^T dz, masked by g > 0
        // (each layer stores its own dY -- the B operand it holds -- slice by slice)
        B bdg[NB128];
        {
            unsigned mk[2];
            SP_BWD_LAYER(9, 0, bdz, masked_to(mk, bdg), masks_of(load_mask(SB_G), mk, NM4{}), store_slice(GB_DZ, 32, 0, NST_16{}, bdz));
        }

        // ---- rgb layer 0 (283 -> 128), transposed: [d feat | d view] = R0^T dg
        B dyA[NB256 + 1], dyB[NB256 + 1];
        {
            unsigned mk[4];
            SP_BWD_LAYER(8, 0, bdg, masked_to(mk, dyA), masks_of(load_mask(SB_FV), mk, NM8{}), store_slice(GB_DG, 128, 0, NST_128{}, bdg));
        }
        if constexpr (POSE) {
            // view-encoding gradient of this sample: 16 slots per lane half, fp32
            auto epi = [&](auto, const f32x16& acc) {
                if (valid) {
                    float* o = a.dv + row * 32;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 t = {acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                        *(f32x4*)(o + (2 * c + h) * 4) = t;
                    }
                }
            };
            SP_BWD_LAYER(8, 1, bdg, epi, no_pre, no_store);
        }
        // raw-sigma slot: q = 128 on half 0 (first slot of C-row block 8)
        if constexpr (KJ == 8) dyA[NB256] = P::zero();
        P::set(dyA, 128, dsig);
        // DY7 row: 9 blocks of 32 columns; block 8 holds only the sigma slot
        B tail[16 / KJ];
#pragma unroll
        for (int q = 0; q < 16; ++q) P::set(tail, q, q == 0 ? dsig : 0.0f);
        auto store_dy7 = [main = store_slice(GB_DY7, 288, 0, NST_256{}, dyA), tl = store_slice(GB_DY7, 288, 256, NST_16{}, tail)](
                             auto gc, auto ngc) {
            main(gc, ngc);
            tl(gc, ngc);
        };

        // ---- feature layers 7..1 transposed, each masked by the saved input activation
        { unsigned mk[4]; SP_BWD_LAYER(7, 0, dyA, masked_to(mk, dyB), masks_of(load_mask(SB_H6), mk, NM8{}), store_dy7); }
        { unsigned mk[4]; SP_BWD_LAYER(6, 0, dyB, masked_to(mk, dyA), masks_of(load_mask(SB_H5), mk, NM8{}), store_slice(GB_DY6, 256, 0, NST_256{}, dyB)); }
        { unsigned mk[4]; SP_BWD_LAYER(5, 0, dyA, masked_to(mk, dyB), masks_of(load_mask(SB_H4), mk, NM8{}), store_slice(GB_DY5, 256, 0, NST_256{}, dyA)); }
        { unsigned mk[4]; SP_BWD_LAYER(4, 0, dyB, masked_to(mk, dyA), masks_of(load_mask(SB_XS), mk, NM8{}), store_slice(GB_DY4, 256, 0, NST_256{}, dyB)); }

        float* dx0 = (float*)(lds + PIPE_LDS_BYTES) + (wave * 64 + lane) * 32;   // POSE only
        if constexpr (POSE) {
            // skip branch: d x0 (first contribution), parked in LDS until layer 0's arrives
            auto epi = [&](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 t = {acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                    *(f32x4*)(dx0 + 16 * mb + 4 * c) = t;
                }
            };
            SP_BWD_LAYER(4, 1, dyB, epi, no_pre, no_store);
        }
        { unsigned mk[4]; SP_BWD_LAYER(3, 0, dyA, masked_to(mk, dyB), masks_of(load_mask(SB_H2), mk, NM8{}), store_slice(GB_DY3, 256, 0, NST_256{}, dyA)); }
        { unsigned mk[4]; SP_BWD_LAYER(2, 0, dyB, masked_to(mk, dyA), masks_of(load_mask(SB_H1), mk, NM8{}), store_slice(GB_DY2, 256, 0, NST_256{}, dyB)); }
        { unsigned mk[4]; SP_BWD_LAYER(1, 0, dyA, masked_to(mk, dyB), masks_of(load_mask(SB_H0), mk, NM8{}), store_slice(GB_DY1, 256, 0, NST_256{}, dyA)); }
        if constexpr (!POSE) {
            // last layer of the chain: nothing left to hide the stores behind
            store_slice(GB_DY0, 256, 0, NST_256{}, dyB)(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        }

        if constexpr (POSE) {
            auto epi = [&](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 t = *(f32x4*)(dx0 + 16 * mb + 4 * c);
                    t[0] += acc[4 * c]; t[1] += acc[4 * c + 1]; t[2] += acc[4 * c + 2]; t[3] += acc[4 * c + 3];
                    *(f32x4*)(dx0 + 16 * mb + 4 * c) = t;
                }
            };
            SP_BWD_LAYER(0, 0, dyB, epi, no_pre, store_slice(GB_DY0, 256, 0, NST_256{}, dyB));

            // positional-encoding backward for this lane half's 15 arguments + raw coords
            const int64_t ray = rowc / a.nsamp;
            const float tt = a.t[rowc];
            const float px = __fadd_rn(a.center[ray * 3 + 0], __fmul_rn(a.dir[ray * 3 + 0], tt));
            const float py = __fadd_rn(a.center[ray * 3 + 1], __fmul_rn(a.dir[ray * 3 + 1], tt));
            const float pz = __fadd_rn(a.center[ray * 3 + 2], __fmul_rn(a.dir[ray * 3 + 2], tt));
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll 1
            for (int i = 0; i < 15; ++i) {
                const int arg = 15 * h + i;
                const int coord = arg >= 20 ? 2 : arg >= 10 ? 1 : 0;
                const int k = arg - 10 * coord;
                const float pv = coord == 0 ? px : coord == 1 ? py : pz;
                const float fr = ldexpf(3.14159274101257324219f, k);
                float s, c;
                sincosf(__fmul_rn(pv, fr), &s, &c);
                const float gq = c2f[k] * fr * (c * dx0[2 * i] - s * dx0[2 * i + 1]);
                g0 += coord == 0 ? gq : 0.f; g1 += coord == 1 ? gq : 0.f; g2 += coord == 2 ? gq : 0.f;
            }
            if (h == 0) { g0 += dx0[30]; g1 += dx0[31]; } else { g2 += dx0[30]; }
            g0 += __shfl_xor(g0, 32); g1 += __shfl_xor(g1, 32); g2 += __shfl_xor(g2, 32);
            if (valid && h == 0) { a.dp[row * 3] = g0; a.dp[row * 3 + 1] = g1; a.dp[row * 3 + 2] = g2; }
        }
#undef SP_BWD_LAYER
    }
    pipe.drain();
}

int launch_mlp_bwd(int prec, bool pose, const MlpBwdArgs& a, int grid, hipStream_t stream) {
    if (a.rows <= 0) return 0;
#define SP_LAUNCH(PR, PO) \
    hipLaunchKernelGGL((mlp_bwd_kernel<PR, PO>), dim3(grid), dim3(Policy<PR>::NWAVES * 64), 0, stream, a)
    if (prec == PREC_BF16) { if (pose) SP_LAUNCH(PREC_BF16, true); else SP_LAUNCH(PREC_BF16, false); }
    else if (prec == PREC_FP32) { if (pose) SP_LAUNCH(PREC_FP32, true); else SP_LAUNCH(PREC_FP32, false); }
    else if (prec == PREC_X3) {
#ifdef SP_X3_DGRAD_FULL
        if (pose) SP_LAUNCH(PREC_X3, true); else SP_LAUNCH(PREC_X3, false);
#else
        // weights head + tail, propagated gradient in bf16 (mlp_dev.h PolicyX3Dgrad): the caller sized the grid
        // for 128-row tiles, the kernel strides over 256-row tiles
        if (pose) hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, true, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
        else hipLaunchKernelGGL((mlp_bwd_kernel<PREC_X3, false, PolicyX3Dgrad>), dim3(grid), dim3(PolicyX3Dgrad::NWAVES * 64), 0, stream, a);
#endif
    }
    else return 1;
#undef SP_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace sparf
