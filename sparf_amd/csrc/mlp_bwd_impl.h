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
//
// This file is the body of two translation units: mlp_bwd.hip (plane areas, every precision) and mlp_bwd_q8.hip (8-bit areas).
#pragma once
#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

#ifdef SP_PROF
// wave-time accounting of wave 0 of workgroup 0 (mlp_dev.h Prof; tools/kernel_bench.py prints it): 0 barrier wait, 1 weight-DMA issue,
// 2 LDS fragments + MFMA issue (with the deferred epilogue units, where there are any), 3 exposed epilogue, 4 mask loads + gradient
// stores + accumulator clear, 5 end of tile (pose: encoding backward), 6 tile inputs
static __device__ unsigned long long g_prof_bwd[10];
#endif

// acc += W_l^T[m-group g of segment S] * dY, over all K parts.
// `pre(g)` runs right after the group's first chunk barrier (group 0 loads the layer's ReLU
// mask words there), then `store(g, ngroups)` issues this group's slice of the layer's dY
// stores: a short burst while the wave waits for its first LDS fragments (mlp_dev.h).
// SP_BWD_SPREAD: the next chunk's weight-DMA pieces are issued one at a time between the MFMAs of the current chunk (mlp_dev.h
// SpreadFetch) instead of as a burst behind the chunk barrier, where all eight waves queue up at the CU's one vector-memory
// port (~570 cycles per 32 KiB chunk at the 58 B/clk an LDS-DMA stream reaches) with the matrix pipe idle.
// 0 = burst everywhere (rounds 1-4), 1 = spread in the kernels WITHOUT pose gradients (default), 2 = spread everywhere.
// Same-box A/B, 786 432 rows, two repetitions (profiles/r05_kernel_ab_spread.log): dgrad bf16 0.916 / 0.908 -> 0.873 / 0.868 ms (-4.5 %),
// bf16x3 1.381 / 1.350 -> 1.339 / 1.333 (-2 %), bf16x3 with 8-bit areas 1.311 / 1.305 -> 1.275 / 1.289 (-2 %); the pose variants, which sit
// at the 256-VGPR limit, spill 9-10 registers with it and measure 1.470 / 1.453 -> 1.481 / 1.469: they keep the burst.
#ifndef SP_BWD_SPREAD
#define SP_BWD_SPREAD 1
#endif
// SP_BWD_DEFER: 1 = the kernels with bf16 gradient operands double-buffer their accumulators and issue a group's mask epilogue behind the
// next group's MFMAs (bwd_layer_deferred below; default), 0 = group by group (rounds 1-4).
#ifndef SP_BWD_DEFER
#define SP_BWD_DEFER 1
#endif
#ifndef SP_BWD_STAGGER
#define SP_BWD_STAGGER 0
#endif
// timing probes (WRONG RESULTS; tools/evidence.sh dgradprobes): SP_PROBE_NO_DMA = the weight chunks are fetched once, every later chunk
// re-reads what the first left in LDS (mlp_dev.h SP_PROBE_NBYTES); SP_PROBE_NO_STORES = no gradient-area stores; SP_PROBE_NO_BARRIER = no
// chunk barriers (with NO_DMA).  What is left of a wave's barrier waits without them is the SIMD's other wave using the matrix pipe.
template <class P, int L, int S, int GI, int NG, bool POSE, int NMB, class Pipe, class Pre, class Store>
SP_DEV void bwd_group(Pipe& pipe, int lane, const typename P::B* dy, f32x16 (&acc)[P::G], Pre&& pre, Store&& store) {
    constexpr int PREC = P::PREC;
    static_for<bwd_nparts(PREC, L)>([&](auto pc) {
        constexpr int part = decltype(pc)::value;
        constexpr int id = bwd_chunk_id(PREC, L, S, GI, part);
        constexpr Chunk cur = bwd_chunk(PREC, id);
        constexpr int nxt = bwd_next_id(PREC, id, POSE);
        constexpr int noff = (int)bwd_chunk_off(PREC, nxt);
        constexpr int nbytes = SP_PROBE_NBYTES(chunk_bytes(PREC, bwd_chunk(PREC, nxt)));
        const char* ch = pipe.template acquire<noff, nbytes>();
        if constexpr (part == 0) {
            pre(std::integral_constant<int, GI>{});        // accumulators not live yet
            store(std::integral_constant<int, GI>{}, std::integral_constant<int, NG>{});
            zero_acc<P, NMB>(acc);
        }
        SP_LAP(pipe.prof, 4);
        mma_chunk<P, cur.nmb, cur.nks>(acc, dy + cur.ks0, ch, lane, SpreadFetch<Pipe, noff, nbytes, cur.nmb, P::NPART>{pipe});
        SP_LAP(pipe.prof, 2);
    });
}

// Deferred mask epilogue (SP_BWD_DEFER, the default since round 5; every kernel with bf16 gradient operands): the accumulators are
// double-buffered and group g-1's epilogue -- BWD_EPI_STAGES units per register pair, `epi(mb, pair, stage, acc)` -- is issued one
// unit per gap behind group g's MFMAs (mlp_dev.h DeferredEpi; with two partial products per k-step the units take the first
// part's gaps, the fragment reads and DMA pieces the second's).  Only the segment's last group keeps an exposed epilogue.
// Round 4 had ruled this out for the 8-wave kernels by register count (244-254 of 256 in use); compiled, it needs 238-252 and no
// scratch (tools/kernel_meta.sh) -- the epilogue's temporaries no longer overlap a full group of live fragments.
// Same box, 786 432 rows, two repetitions, results bit-identical (profiles/r05_kernel_ab_dgrad_defer{,_bf16}.log, r05_dgrad_defer_digests.log):
//   bf16x3  dgrad 1.338-1.357 -> 1.307-1.318 ms   8-bit areas 1.302-1.304 -> 1.260-1.269   with pose gradients 1.469-1.486 -> 1.458-1.467
//   bf16          0.880-0.897 -> 0.852                        0.842-0.847 -> 0.811-0.814                       1.010-1.024 -> 0.977-0.991
// i.e. -2.5 ... -4 %, although the exposed epilogues were 9.3 % of a wave's cycles (wave-time accounting, profiles/r05_dgrad_lap_table.log:
// epilogue 9.3 % -> 1.8 %, the wave's total -7 %): with two waves per SIMD the other wave's MFMAs already filled most of that time.
// The 4-wave variant (-DSP_X3_DGRAD_WAVES=4: one wave per SIMD, 128-row tiles, the forward's geometry) with the same deferral:
// 1.49-1.59 ms, 12-17 % SLOWER than the 8-wave kernel -- twice the weight-stream traffic per row and nobody to issue while the
// wave queues at the vector-memory port.  Kept as a build flag.
// What the timing probes below say about the rest (pose kernel, same log): no stores 1.551 -> 1.267 ms, no weight DMA -> 1.385, neither
// -> 1.158, and without the chunk barriers as well 1.136: the barriers themselves are 2 %, the 36 % "barrier" share of a wave's cycles
// is the SIMD's other wave using the matrix pipe, and what the kernel pays for is its vector-memory traffic -- the gradient stores
// (3.5 GB per launch) ~20 %, the weight stream ~10 % -- on top of an issue-bound floor at 74 % of the matrix pipe.
enum { BWD_EPI_STAGES = 2 };
template <class P, int L, int NMB> SP_DEV constexpr int bwd_group_mfmas_before(int part_end) {
    int n = 0;
    for (int p = 0; p < bwd_nparts(P::PREC, L); ++p) {
        if (p == part_end) return n;
        n += bwd_part_nks(P::PREC, L, p) * NMB * P::NPART;
    }
    return n;
}
template <class P, int L, int S, bool POSE, class Pipe, class Epi, class Pre, class Store>
SP_DEV void bwd_layer_deferred(Pipe& pipe, int lane, const typename P::B* dy, Epi&& epi, Pre&& pre, Store&& store) {
    constexpr int PREC = P::PREC, G = P::G;
    constexpr int NG = bwd_seg_ngroups(PREC, L, S), TOT = vk_width(layer_seg_kind(L, S)) / 32;
    f32x16 accs[2][G];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value, mb0 = g * G;
        constexpr int nmb = (TOT - mb0) < G ? (TOT - mb0) : G;
        constexpr int nmb_prev = g > 0 ? G : 0;                 // every group but the last is full
        constexpr int ntot = bwd_group_mfmas_before<P, L, nmb>(-1);
        f32x16 (&acc)[G] = accs[g & 1];
        static_for<bwd_nparts(PREC, L)>([&](auto pc) {
            constexpr int part = decltype(pc)::value;
            constexpr int id = bwd_chunk_id(PREC, L, S, g, part);
            constexpr Chunk cur = bwd_chunk(PREC, id);
            constexpr int nxt = bwd_next_id(PREC, id, POSE);
            constexpr int noff = (int)bwd_chunk_off(PREC, nxt);
            constexpr int nbytes = SP_PROBE_NBYTES(chunk_bytes(PREC, bwd_chunk(PREC, nxt)));
            const char* ch = pipe.template acquire<noff, nbytes>();
            // SP_BWD_STAGGER (experiment, off): the two waves of a SIMD (w, w + 4) issue their store bursts behind DIFFERENT chunk barriers of
            // the group, so that one of them keeps the matrix pipe fed while the other queues at the CU's vector-memory port.  Bit-identical;
            // bf16x3 dgrad 1.332-1.336 -> 1.321-1.338 ms, 8-bit areas 1.285 -> 1.256-1.265, with pose gradients 1.483-1.486 -> 1.477-1.493 and 12
            // spilled registers (profiles/r05_kernel_ab_dgrad_final.log): the stores' price is not their collision at the port.
            constexpr bool STAG = SP_BWD_STAGGER && bwd_nparts(PREC, L) >= 2 && P::NWAVES == 8;
            if constexpr (part == 0) {
                pre(gc);
                if (!STAG || pipe.wave < 4) store(gc, std::integral_constant<int, NG>{});
                zero_acc<P, nmb>(acc);
            }
            if constexpr (part == 1 && STAG) {
                if (pipe.wave >= 4) store(gc, std::integral_constant<int, NG>{});
            }
            SP_LAP(pipe.prof, 4);
            if constexpr (g > 0) {
                constexpr int base = bwd_group_mfmas_before<P, L, nmb>(part);
                mma_chunk<P, cur.nmb, cur.nks>(acc, dy + cur.ks0, ch, lane,
                                               DeferredEpi<P, Pipe, std::remove_reference_t<Epi>, nmb_prev, mb0 - G, base, ntot, noff, nbytes, nmb, BWD_EPI_STAGES>{
                                                   pipe, epi, accs[(g & 1) ^ 1]});
            } else {
                mma_chunk<P, cur.nmb, cur.nks>(acc, dy + cur.ks0, ch, lane, SpreadFetch<Pipe, noff, nbytes, cur.nmb, P::NPART>{pipe});
            }
            SP_LAP(pipe.prof, 2);
        });
        if constexpr (g == NG - 1) {
            static_for<nmb * 8 * BWD_EPI_STAGES>([&](auto uc) {
                constexpr int u = decltype(uc)::value, p = u / BWD_EPI_STAGES;
                epi(std::integral_constant<int, mb0 + p / 8>{}, std::integral_constant<int, p % 8>{}, std::integral_constant<int, u % BWD_EPI_STAGES>{}, acc[p / 8]);
            });
            SP_LAP(pipe.prof, 3);
        }
    });
}

// Q8: the save area (mask words) and the gradient area are in the 8-bit format (layout.h AREA_Q8): every dY vector leaves as
// signed 8-bit integers with one step per row (mlp_dev.h "8-bit saves")
template <int PREC, bool POSE, class P = Policy<PREC>, bool Q8 = false>
__global__ void __launch_bounds__(P::NWAVES * 64) mlp_bwd_kernel(MlpBwdArgs a) {
    typedef typename P::B B;
    constexpr int AF = area_format(PREC, Q8);
    static_assert(!Q8 || sizeof(B) == 16, "8-bit gradient area: bf16 gradient operands only");
    constexpr int KJ = P::KJ, CH = P::CH, NW = P::NWAVES, G = P::G;
    constexpr int NB256 = 128 / KJ, NB128 = 64 / KJ;

    constexpr int DX_BYTES = POSE ? NW * 64 * 32 * 4 : 0;
    __shared__ __attribute__((aligned(16))) char lds[PIPE_LDS_BYTES + DX_BYTES + 64];          // weight pipe | POSE: d x0 stash | c2f band weights

    const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int64_t BWD_OFF = packed_bwd_off(PREC);
    constexpr unsigned BWD_BYTES = (unsigned)bwd_stream_bytes(PREC);
    constexpr int C0_BYTES = chunk_bytes(PREC, bwd_chunk(PREC, 0));
    float* c2f = (float*)(lds + PIPE_LDS_BYTES + DX_BYTES);      // the ten position-band weights of the pass, read per lane by the encoding backward
    if (POSE && threadIdx.x < 10) c2f[threadIdx.x] = a.c2f[threadIdx.x];             // (visible after the first chunk barrier)

    WeightPipe<NW, (SP_BWD_SPREAD == 2 || (SP_BWD_SPREAD == 1 && (!POSE || NW == 4)))> pipe;
    pipe.init(a.packed + BWD_OFF, BWD_BYTES, lds);
    pipe.prime(0, C0_BYTES);
#ifdef SP_PROBE_NO_DMA
    pipe.fetch(0, C0_BYTES, 1u);
#endif

    const int64_t rows = a.rows;                // active rows: [row_begin, rows)
    const int tile_rows = NW * 32;
    const int64_t ntiles = (rows - a.row_begin + tile_rows - 1) / tile_rows;
    const int64_t tile32_0 = a.row_begin >> 5, area_tiles = ntiles32(a.rows_total);
    const int lvo = lane_voff(n, h);            // lane part of every gradient-store address (mlp_dev.h)

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // tile-block-major areas (layout.h): this wave's 32 rows form tile `tile32` (wave-uniform)
        const int64_t tile32 = tile32_0 + tile * NW + wave;
        const int64_t row = tile32 * 32 + n;
        const bool valid = row < rows;
        const int64_t rowc = valid ? row : rows - 1;
        // this wave's tile blocks of the save area (mask words) and of the gradient area (layout.h)
        // (a range that starts at row_begin > 0 is not aligned to the workgroup tile any more: the waves past its end may own
        // tiles beyond the padded areas -- they keep running for the barriers, through zero-sized descriptors)
        const bool in_area = tile32 < area_tiles;
        const __amdgpu_buffer_rsrc_t srs = tile_rsrc<P>(a.save, tile32, save_tile_bytes(AF), in_area);
        const __amdgpu_buffer_rsrc_t grs = tile_rsrc<P>(a.grad, tile32, grad_tile_bytes(AF), in_area);
        // 16-byte chunks [0, NST) of gradient vector v -> columns COL0.. of grad buffer GB;
        // accumulator group g of ng stores its share
        // (8-bit format: group 0 takes max |dY| over the whole vector -- complete in registers, it is this layer's B operand --
        // sets the quantiser factor `qf` and stores the row's step; every group quantises its share of 16-slot blocks)
        float qf_a = 0.0f, qf_b = 0.0f;
        auto store_slice = [&](auto gbc, auto col0c, auto nstc, const B* v, float& qf) {
            return [&grs, &qf, lvo, n, v](auto gc, auto ngc) {
#ifdef SP_PROBE_NO_STORES
                return;
#endif
                constexpr int NST = decltype(nstc)::value, g = decltype(gc)::value, ng = decltype(ngc)::value;
                constexpr int gb = decltype(gbc)::value, col0 = decltype(col0c)::value;
                if constexpr (Q8) {
                    constexpr int N16 = NST / 2, b0 = g * N16 / ng, b1 = (g + 1) * N16 / ng;
                    constexpr int part = col0 >= 256 ? 1 : 0;
                    constexpr int BASE = grad_buf_tile_off(AF, gb) + (col0 / 32) * 1024;
                    if constexpr (g == 0) {
                        const float amax = q8_amax<false, NST>(v);
                        qf = q8_factor(amax);
                        q8_store_step<grad_step_tile_off(gb, part)>(grs, n, amax);
                    }
                    if constexpr (b1 > b0) static_for<b1 - b0>([&](auto cc) { q8_store16<BASE, b0 + decltype(cc)::value>(grs, lvo, v, qf); });
                } else {
                    constexpr int c0 = g * NST / ng, c1 = (g + 1) * NST / ng;
                    constexpr int BASE = grad_buf_tile_off(AF, gb) + (col0 / CH) * 512;
                    if constexpr (c1 > c0)
                        static_for<c1 - c0>([&](auto cc) { bstore_chunk<P, BASE, c0 + decltype(cc)::value, (int)grad_plane_tile_bytes(AF)>(grs, lvo, v); });
                }
            };
        };
        typedef std::integral_constant<int, 128 / CH> NST_256;
        typedef std::integral_constant<int, 64 / CH> NST_128;
        typedef std::integral_constant<int, 16 / CH> NST_16;
        typedef std::integral_constant<int, 0> C0;
        typedef std::integral_constant<int, 256> C256;
#define SP_ID(b) std::integral_constant<int, b>{}
        // group 0 first loads the layer's ReLU mask words (layout.h "ReLU masks": four 32-bit FIFO words per lane,
        // written by the forward kernel as one 16-byte store) -- BEFORE the layer's stores, so that waiting for
        // them later does not wait for these stores (vmcnt retires in issue order)
        auto masks_of = [&](auto sbc, unsigned* mk, auto nmc) {
            return [&srs, lane, mk](auto gc) {
                if constexpr (decltype(gc)::value == 0) {
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(srs, lane * 16, save_mask_tile_off(AF, decltype(sbc)::value), SP_SAVE_AUX);
#pragma unroll
                    for (int p = 0; p < decltype(nmc)::value / 2; ++p) mk[p] = w[p];
                }
            };
        };
        // epilogue: dy_prev[q] = acc * [saved activation > 0].  The element's bit is popped from the HIGH end of
        // the lane's FIFO word: v_add_co_u32 word, <sgpr pair>, word, word leaves the popped bits of all 64 lanes as a
        // lane mask in an SGPR pair, which the select consumes directly (v_cndmask_b32 y, 0, acc, <sgpr pair>): two
        // instructions where shift + and + compare + select took four.  Only the add is inline asm: the select is
        // the compiler's, because it READS AN MFMA RESULT and the wait states between an MFMA and a VALU read of its
        // destination are software-managed on gfx950 -- the hazard recogniser inserts them for its own instructions,
        // not inside asm statements.  Elements are popped in the order the forward pushed them: m-block ascending,
        // register ascending.
        auto masked_to = [&](unsigned* mk, B* out) {
            return [mk, out](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
                float prev = 0.0f;       // (an unused operand of the next pop: keeps pop r+1 behind select r, i.e. one lane mask live at a time --
                                         //  left free, or chained only pair-wise, the scheduler hoists the pops and spills their SGPR pairs through
                                         //  v_writelane / v_readlane: 1600 extra instructions per tile, measured in the ISA)
                auto pop = [&](int r) {
                    unsigned long long lanes;
                    asm("v_add_co_u32 %0, %1, %0, %0" : "+v"(mk[mb / 2]), "=s"(lanes) : "v"(prev));
                    prev = __builtin_amdgcn_inverse_ballot_w64(lanes) ? acc[r] : 0.0f;
                    return prev;
                };
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float y0 = pop(r), y1 = pop(r + 1);
                    const int q0 = 16 * mb + r;
                    if constexpr (sizeof(B) == 16) {           // bf16 operands: the pair leaves in ONE v_cvt_pk_bf16_f32
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        const bf16x2_t hp = {(__bf16)y0, (__bf16)y1};
                        u32x4 t = __builtin_bit_cast(u32x4, out[q0 >> 3]);
                        t[(q0 & 7) >> 1] = __builtin_bit_cast(unsigned, hp);
                        out[q0 >> 3] = __builtin_bit_cast(bf16x8, t);
                    } else {
                        P::set(out, q0, y0);
                        P::set(out, q0 + 1, y1);
                    }
                }
            };
        };
        // the same epilogue cut into units for bwd_layer_deferred: stage 0 / 1 = pop + select of element 0 / 1 of the pair, stage 1 also packs
        // the pair into the next B operand
        float e_prev = 0.0f, e_y0 = 0.0f;
        auto masked_units = [&](unsigned* mk, B* out) {
            return [mk, out, &e_prev, &e_y0](auto mbc, auto pairc, auto stagec, const f32x16& acc, auto... deferred) {
                constexpr int mb = decltype(mbc)::value, pr = decltype(pairc)::value, st = decltype(stagec)::value, r = 2 * pr + st;
                unsigned long long lanes;
                asm("v_add_co_u32 %0, %1, %0, %0" : "+v"(mk[mb / 2]), "=s"(lanes) : "v"(e_prev));
                const float y = __builtin_amdgcn_inverse_ballot_w64(lanes) ? acc[r] : 0.0f;
                e_prev = y;
                if constexpr (st == 0) {
                    e_y0 = y;
                } else if constexpr (sizeof(B) == 16) {                  // (deferred dgrad epilogue: bf16 gradient operands only)
                    constexpr int q0 = 16 * mb + 2 * pr;
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const bf16x2_t hp = {(__bf16)e_y0, (__bf16)y};
                    u32x4 t = __builtin_bit_cast(u32x4, out[q0 >> 3]);
                    t[(q0 & 7) >> 1] = __builtin_bit_cast(unsigned, hp);
                    out[q0 >> 3] = __builtin_bit_cast(bf16x8, t);
                }
            };
        };
        constexpr bool DEFER = SP_BWD_DEFER && sizeof(B) == 16;
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
            SP_LAP(pipe.prof, 3);                                                            \
        })

        // a masked layer: deferred (4-wave kernels) or group by group
#define SP_BWD_MASKED(L, DY, MK, OUT, PRE, STORE)                                               \
        do {                                                                                    \
            if constexpr (DEFER) bwd_layer_deferred<P, L, 0, POSE>(pipe, lane, DY, masked_units(MK, OUT), PRE, STORE); \
            else SP_BWD_LAYER(L, 0, DY, masked_to(MK, OUT), PRE, STORE);                        \
        } while (0)

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
        SP_LAP(pipe.prof, 6);

        // ---- rgb layer 1 (128 -> 3), transposed: dg = R1^T dz, masked by g > 0
        // (each layer stores its own dY -- the B operand it holds -- slice by slice)
        B bdg[NB128];
        {
            unsigned mk[2];
            SP_BWD_MASKED(9, bdz, mk, bdg, masks_of(SP_ID(SB_G), mk, NM4{}), store_slice(SP_ID(GB_DZ), C0{}, NST_16{}, bdz, qf_a));
        }

        // ---- rgb layer 0 (283 -> 128), transposed: [d feat | d view] = R0^T dg
        B dyA[NB256 + 1], dyB[NB256 + 1];
        {
            unsigned mk[4];
            SP_BWD_MASKED(8, bdg, mk, dyA, masks_of(SP_ID(SB_FV), mk, NM8{}), store_slice(SP_ID(GB_DG), C0{}, NST_128{}, bdg, qf_a));
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
        auto store_dy7 = [main = store_slice(SP_ID(GB_DY7), C0{}, NST_256{}, dyA, qf_a), tl = store_slice(SP_ID(GB_DY7), C256{}, NST_16{}, tail, qf_b)](
                             auto gc, auto ngc) {
            main(gc, ngc);
            tl(gc, ngc);
        };

        // ---- feature layers 7..1 transposed, each masked by the saved input activation
        { unsigned mk[4]; SP_BWD_MASKED(7, dyA, mk, dyB, masks_of(SP_ID(SB_H6), mk, NM8{}), store_dy7); }
        { unsigned mk[4]; SP_BWD_MASKED(6, dyB, mk, dyA, masks_of(SP_ID(SB_H5), mk, NM8{}), store_slice(SP_ID(GB_DY6), C0{}, NST_256{}, dyB, qf_a)); }
        { unsigned mk[4]; SP_BWD_MASKED(5, dyA, mk, dyB, masks_of(SP_ID(SB_H4), mk, NM8{}), store_slice(SP_ID(GB_DY5), C0{}, NST_256{}, dyA, qf_a)); }
        { unsigned mk[4]; SP_BWD_MASKED(4, dyB, mk, dyA, masks_of(SP_ID(SB_XS), mk, NM8{}), store_slice(SP_ID(GB_DY4), C0{}, NST_256{}, dyB, qf_a)); }

        // POSE only: d x0 of this lane, 32 floats, as [4-float chunk 0..7][lane][4] (lane-contiguous 16-byte
        // slots: conflict-free ds_*_b128; round 2 kept a lane's 32 floats contiguous, a 128-byte lane stride that
        // put a whole lane group on one bank -- the 16 % LDS-conflict share of the pose dgrad's PMC profile)
        float* dxw = (float*)(lds + PIPE_LDS_BYTES) + wave * (64 * 32);
        auto dx0c = [&](int chunk) { return (f32x4*)(dxw + (chunk * 64 + lane) * 4); };
        if constexpr (POSE) {
            // skip branch: d x0 (first contribution), parked in LDS until layer 0's arrives
            auto epi = [&](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 t = {acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
                    *dx0c(4 * mb + c) = t;
                }
            };
            SP_BWD_LAYER(4, 1, dyB, epi, no_pre, no_store);
        }
        { unsigned mk[4]; SP_BWD_MASKED(3, dyA, mk, dyB, masks_of(SP_ID(SB_H2), mk, NM8{}), store_slice(SP_ID(GB_DY3), C0{}, NST_256{}, dyA, qf_a)); }
        { unsigned mk[4]; SP_BWD_MASKED(2, dyB, mk, dyA, masks_of(SP_ID(SB_H1), mk, NM8{}), store_slice(SP_ID(GB_DY2), C0{}, NST_256{}, dyB, qf_a)); }
        { unsigned mk[4]; SP_BWD_MASKED(1, dyA, mk, dyB, masks_of(SP_ID(SB_H0), mk, NM8{}), store_slice(SP_ID(GB_DY1), C0{}, NST_256{}, dyA, qf_a)); }
        if constexpr (!POSE) {
            // last layer of the chain: nothing left to hide the stores behind
            store_slice(SP_ID(GB_DY0), C0{}, NST_256{}, dyB, qf_a)(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        }

        if constexpr (POSE) {
            auto epi = [&](auto mbc, const f32x16& acc) {
                constexpr int mb = decltype(mbc)::value;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 t = *dx0c(4 * mb + c);
                    t[0] += acc[4 * c]; t[1] += acc[4 * c + 1]; t[2] += acc[4 * c + 2]; t[3] += acc[4 * c + 3];
                    *dx0c(4 * mb + c) = t;
                }
            };
            SP_BWD_LAYER(0, 0, dyB, epi, no_pre, store_slice(SP_ID(GB_DY0), C0{}, NST_256{}, dyB, qf_a));

            // positional-encoding backward for this lane half's 15 arguments + raw coords
            const int64_t ray = rowc / a.nsamp;
            const float tt = a.t[rowc];
            const float px = __fadd_rn(a.center[ray * 3 + 0], __fmul_rn(a.dir[ray * 3 + 0], tt));
            const float py = __fadd_rn(a.center[ray * 3 + 1], __fmul_rn(a.dir[ray * 3 + 1], tt));
            const float pz = __fadd_rn(a.center[ray * 3 + 2], __fmul_rn(a.dir[ray * 3 + 2], tt));
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            // half 0: x:k0..9, y:k0..4; half 1: y:k5..9, z:k0..9 (the forward's argument order): gA / gB = the lane's first / second coordinate
            const float pvA = h ? py : px, pvB = h ? pz : py;
            const int split = h ? 5 : 10, kA0 = h ? 5 : 0;
            float gA = 0.f, gB = 0.f;
#pragma unroll 1
            for (int i = 0; i < 15; ++i) {
                const bool first = i < split;
                const int k = first ? kA0 + i : i - split;
                const float fr = ldexpf(3.14159274101257324219f, k);
                float s, c;
                sincosf(__fmul_rn(first ? pvA : pvB, fr), &s, &c);
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 dsc = *(const f32x2*)((const float*)dx0c(i >> 1) + 2 * (i & 1));        // d sin, d cos slots 2i, 2i+1
                const float gq = c2f[k] * fr * (c * dsc[0] - s * dsc[1]);
                gA += first ? gq : 0.f; gB += first ? 0.f : gq;
            }
            if (h == 0) { g0 = gA; g1 = gB; } else { g1 = gA; g2 = gB; }
            { const f32x4 raw = *dx0c(7); if (h == 0) { g0 += raw[2]; g1 += raw[3]; } else { g2 += raw[2]; } }      // slots 30, 31
            g0 += __shfl_xor(g0, 32); g1 += __shfl_xor(g1, 32); g2 += __shfl_xor(g2, 32);
            if (valid && h == 0) { a.dp[row * 3] = g0; a.dp[row * 3 + 1] = g1; a.dp[row * 3 + 2] = g2; }
        }
#undef SP_BWD_MASKED
#undef SP_BWD_LAYER
#undef SP_ID
        SP_LAP(pipe.prof, 5);
    }
    pipe.drain();
#ifdef SP_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 10; ++i) g_prof_bwd[i] = pipe.prof.acc[i];
#endif
}

}  // namespace sparf
