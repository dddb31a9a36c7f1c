// Weight gradients: dW_l[out][in] = sum over sample rows of dY_l[row][out] * X_l[row][in]
// for the ten layers, as ten split-K MFMA GEMMs ("jobs", layout.h) over the activations
// saved by the forward kernel (X) and the pre-activation gradients written by the dgrad
// kernel (dY).  The contraction runs over rows, so both operands are transposed on their
// way LDS -> registers.  Each workgroup owns one (job, row-range) pair and writes an fp32
// partial [32*MB][32*NB] matrix (+ the bias gradient = column sums of dY); a second kernel
// sums the partials over the splits and scatters them into nn.Linear [out][in] order.
// Deterministic: no atomics.
//
// HBM-bound by design: 2*(M+N)*sizeof(act) bytes per row against 2*M*N flops.
#include <type_traits>

#include "kernels.h"
#include "mlp_dev.h"

namespace sparf {

enum { WG_THREADS = 512, WGRAD_LDS_BYTES = 160 * 1024 };
// streamed-once operands: non-temporal LDS-DMA (MI355X_MICROARCH.md "nt-weights": lands ~18 % sooner)
#ifndef SP_WG_NT
#define SP_WG_NT " nt"
#endif
// SP_WG_SPREAD = 1: the refill of the operand ring (the DMA pieces of tile t + DEPTH) is issued piece by piece BETWEEN the MFMAs of
// tile t instead of as a burst behind the tile's barrier.  Behind the barrier all eight waves queue up at the CU's one vector-memory
// port: 32 KiB per tile at the ~58 B/clk an LDS-DMA stream reaches (tools/probes/vmem_probe.hip) is ~570 cycles in which no wave
// issues an MFMA, next to the 1 024 cycles the tile's MFMAs occupy the matrix pipe -- the 1 900 cycles per tile round 4 measured.
// MEASURED AND NOT ADOPTED (round 5, same-box A/B, 786 432 rows, two repetitions, profiles/r05_kernel_ab_spread.log): bf16 planes
// 1.353 / 1.326 -> 1.357 / 1.329 ms, 8-bit operands 1.371 / 1.377 -> 1.371 / 1.350 ms: nothing.  The flag stays for the next attempt.
// Also built and measured in round 5, and removed again (commit 17caff4 has the code; profiles/r05_kernel_ab_wgrad_pipe.log): the tile loop
// as ONE software pipeline across tiles -- the barrier that publishes tile t+1 in the MIDDLE of tile t's MFMA stream, the fragment ring
// and the fixed operand's fragments of tile t+1 read behind it during the second half of tile t's MFMAs, the refill spread over them,
// for 8-bit operands a third bf16 image so that tile t+1 is converted while tile t is multiplied: bit-identical, no spills, and
// bf16 planes 1.319 / 1.314 -> 1.315 / 1.325 ms (nothing), 8-bit operands 1.352 / 1.333 -> 1.486 / 1.443 ms (SLOWER: the third image
// costs a tile of prefetch depth, the conversion sits in the MFMA stream).  So neither the refill burst nor the per-tile barrier with
// its cold fragment ring is what the loop waits for.  PMC (profiles/r05_pmc_bf16x3.json, r05_pmc_bf16x3+q8.json): matrix pipe busy 31 % / 30 %
// (planes / 8-bit), LDS array busy 19.6 % / 29.8 %, no bank conflicts: no single resource is the bound.  With planes the HBM stream is
// (5.4-5.8 TB/s = 85-90 % of the 6.3-6.8 TB/s an LDS-DMA stream reaches on this chip); with 8-bit operands a workgroup is one
// dependency chain per tile (land -> convert -> barrier -> fragments -> MFMAs) at one workgroup per CU, and its latencies add up.
// What would overlap them is two independent workgroups per CU (<= 80 KiB of LDS each), i.e. another kernel geometry.
#ifndef SP_WG_SPREAD
#define SP_WG_SPREAD 0
#endif

// sum of the contraction elements one lane holds in an operand fragment (bias gradient)
template <int PREC> struct WOps;
template <> struct WOps<PREC_BF16> {
    static SP_DEV float fsum(bf16x8 v) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j];
        return s;
    }
};
template <> struct WOps<PREC_FP32> {
    static SP_DEV float fsum(float v) { return v; }
};

// ------------------------------------------------------------------ LDS-DMA pipeline
// The saved buffers are tile-major ([tile32][16-byte chunk][row&31][8 elements], layout.h),
// so a 32-row x C-column tile is one contiguous block and its LDS image is made a straight
// copy of it by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, no staging
// registers, no ds_write) -- except that inside each 512-byte chunk block the 32 row slots
// are XOR-swizzled, slot = row ^ ((chunk & 3) << 2), applied on the per-lane SOURCE offset
// and again on the reads: the four chunks a transposing read pass touches then sit on four
// different 64-byte bank groups.
// Ring of NBUF 32-row tile buffers, DEPTH tiles in flight, one barrier per tile:
//     counted vmcnt (tile t landed) -> s_barrier (visible to all, buffer of t-1 free)
//     -> issue DMA(t+DEPTH) -> MFMA on tile t.
// NPL = 2 (bf16x3): every tile carries a head plane and a tail plane of both operands
// ([dY hi][X hi][dY lo][X lo] in LDS) and a block product is three MFMAs.
// ROWS = rows per ring slot: 32 (a whole layout tile, two MFMA k-steps) or 16 (half a tile, one
// k-step): with two planes a 32-row slot is 64-72 KiB and only two fit in LDS, 16-row slots
// keep a four-slot ring (three in flight).
// EB = 4: fp32 operands (v_mfma_f32_32x32x2_f32, one float per lane and k-step, plain
// ds_read_b32 with the same row-slot swizzle: 2-way bank conflicts, MFMA-bound anyway).
// Q8 (layout.h AREA_Q8, bf16-operand modes): both operands arrive as 8-bit integers with one fp32 step per row and vector, half
// the bytes.  The MFMA wants bf16, so a second stage sits between the DMA and the image the fragments are read from: a 32-row
// tile's 1 KiB pieces (one 32-column block each: lanes 0-31 / 32-63 = the two lane halves' 16 slots of the rows) land in an LDS ring
// QD tiles ahead, straight copies; the wave that fetched a piece reads it back, multiplies it out ((u - 128) * step, one step per
// lane: a lane's 16 bytes belong to ONE row) and writes it as two bf16 chunk blocks into the same swizzled image the DMA path builds
// directly; everything from the transposing reads on is shared.  Two images: a wave that writes tile t+1 has passed the barrier of
// tile t, so every wave is done with tile t-1.  (Staging the pieces in registers instead of the LDS ring spilled: the 8 x 10 job
// alone holds 160 accumulator registers.)
// Measured (786 432 rows, same box, profiles/r04_q8_*.log): 1.35 ms -- the time of the bf16 operands with half their bytes.  Probe
// builds say why: the DMA stream alone 0.79 ms, + the conversion 0.84 ms, + the MFMAs WITHOUT conversion 1.06 ms, the MFMA loop on
// its own (no refills) 0.93 ms = 1 900 cycles per 32-row tile where its 2 x 16 MFMAs per SIMD occupy the matrix pipe for 1 024:
// all eight waves meet at one barrier per tile, so the latencies around it (first fragment reads, pipe drain, barrier skew) are paid
// per tile and nothing of another tile's work overlaps them; removing the fragment reads (0.90 ms) or the barrier (-0.1 ms) does not
// change that.  Behind 7.2 GB of bf16 operands (1.07 ms at the 6.7 TB/s a copy reaches) this hides; behind 3.6 GB it is what is left.
// Two re-schedules were built and measured slower: the conversion cut into four-slot units issued behind the MFMAs of the previous
// tile, step rows fetched once per workgroup one tile ahead, DMA operations spread over the MFMA loop (1.43 ms), and the same with
// the conversion as a phase behind the MFMAs (1.47 ms).  Removed again.
// LDS_BYTES / mb0 (8-bit operands only): a HALF job -- m-blocks [mb0, mb0 + MB) of a job whose dY operand is wider than 32 MB columns --
// inside an 80 KiB LDS budget, so that two workgroups are resident per CU (wgrad_q8h_kernel below)
template <int MB, int NB, int NPL = 1, int ROWS = 32, int EB = 2, bool Q8 = false, int LDS_BYTES = WGRAD_LDS_BYTES>
SP_DEV void wgrad_job_dma(const WgradArgs& a, int job, char* lds, int mb0 = 0) {
    constexpr bool FP32 = EB == 4;
    static_assert(!Q8 || (NPL == 1 && ROWS == 32 && EB == 2), "8-bit operands: one bf16 image of whole layout tiles");
    typedef typename std::conditional<FP32, Policy<PREC_FP32>, Policy<PREC_BF16>>::type P;
    typedef typename std::conditional<FP32, float, bf16x8>::type frag_t;
    static_assert(ROWS == 32 || ROWS == 16, "ring slot = a layout tile or half of one");
    static_assert(!FP32 || (ROWS == 16 && NPL == 1), "fp32 operands: 16-row slots, one plane");
    constexpr int CS = ROWS * 16;                  // LDS bytes of one 16-byte-chunk block (ROWS row slots)
    constexpr int CPP = 1024 / CS;                 // chunk blocks per 1 KiB DMA piece
    constexpr int KSTEPS = FP32 ? ROWS / 2 : ROWS / 16;                   // MFMA k-steps per slot (K = 2 / 16 rows)
    constexpr int64_t WPARTIAL = wpartial_floats();
    constexpr int EPC = 16 / EB;                                           // elements per 16-byte chunk
    constexpr int M = 32 * MB, N = 32 * NB, CM = M / EPC, CN = N / EPC;    // 16-byte chunks per row
    constexpr int CPB = 32 / EPC;                                          // chunk blocks per 32-column block
    constexpr int DY_BYTES = CM * CS, X_BYTES = CN * CS, PLANE_BYTES = DY_BYTES + X_BYTES, BUF_BYTES = NPL * PLANE_BYTES;
    // ring size: 4 buffers / 3 tiles (96 KiB) in flight per CU.  Filling the whole LDS (up to 8
    // buffers for the narrow jobs) measured the same 1.38 ms: the kernel is not latency-bound.
#ifndef SP_WG_NBUF_MAX
#define SP_WG_NBUF_MAX 4
#endif
    constexpr int NBUF = Q8 ? 2 : WGRAD_LDS_BYTES / BUF_BYTES < SP_WG_NBUF_MAX ? WGRAD_LDS_BYTES / BUF_BYTES : SP_WG_NBUF_MAX;
    constexpr int DEPTH = NBUF - 1;
    constexpr int PIECES = BUF_BYTES / 1024;                              // 1 KiB DMA pieces per tile
    constexpr int PPW_HI = (PIECES + 7) / 8, PPW_LO = PIECES / 8, N_HI = PIECES % 8;   // waves < N_HI issue PPW_HI pieces

    const WJob jb = wjob(job);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // tile-block-major areas (layout.h): operand tile (tile32, buffer) = area + tile32 * tile bytes + buffer offset
    constexpr int WPREC = FP32 ? PREC_FP32 : NPL == 2 ? PREC_X3 : PREC_BF16;       // (head planes of bf16x3 = the bf16 image)
    static_assert(NPL == 1 || nplanes_of(PREC_X3) == 2, "two-plane jobs need the two-plane bf16x3 areas");
    constexpr int64_t DY_TILE = NPL == 2 ? grad_tile_bytes(PREC_X3) : FP32 ? grad_tile_bytes(PREC_FP32) : grad_tile_bytes(PREC_BF16);
    constexpr int64_t X_TILE = NPL == 2 ? save_tile_bytes(PREC_X3) : FP32 ? save_tile_bytes(PREC_FP32) : save_tile_bytes(PREC_BF16);
    constexpr int64_t DY_PLANE = grad_plane_tile_bytes(WPREC), X_PLANE = save_plane_tile_bytes(WPREC);   // tail plane inside the tile block
    const char* dy_base = (const char*)a.grad + grad_coloff(jb.gbuf) * 32 * EB;
    const char* x_base = (const char*)a.save + (save_coloff(jb.sbuf) * 32 + (int64_t)jb.xcol0 / EPC * 32 * EPC) * EB;
    (void)dy_base; (void)x_base;

    const int64_t r_begin = a.row_begin + (int64_t)blockIdx.x * a.rows_per_split;
    const int64_t r_end = r_begin + a.rows_per_split < a.rows ? r_begin + a.rows_per_split : a.rows;
    const int ntiles = r_begin < r_end ? (int)((r_end - r_begin + ROWS - 1) / ROWS) : 0;   // last tile ends <= rows_pad

    // DMA source offset of this lane inside a 1 KiB piece = CPP chunk blocks (chunk CPP*q + cip):
    // lane L fills row slot (L % ROWS) of chunk block cip = L / ROWS and must fetch the row
    // slot ^ swizzle(chunk & 3) of that chunk (global chunk blocks are always 32 rows = 512 B)
    const int cip = lane / ROWS, slot = lane % ROWS;
    const int voff_even = (cip * 32 + (slot ^ (((0 + cip) & 3) << 2))) * 16;            // CPP*q = 0 (mod 4)
    const int voff_odd = (cip * 32 + (slot ^ (((CPP + cip) & 3) << 2))) * 16;          // CPP*q = 2 (mod 4): only when CPP == 2

    // piece i (of PPW_HI) of this wave for tile t
    auto issue_piece = [&](int t, int i) {
        char* dst = lds + (t % NBUF) * BUF_BYTES;
        const int64_t tile32 = (r_begin >> 5) + t / (32 / ROWS);
        const int half_off = (t % (32 / ROWS)) * 256;                      // second 16 rows of the layout tile
        {
            const int p = i * 8 + wave;                    // wave-uniform piece id
            if (p < PIECES) {
                const int plane = p / (PLANE_BYTES / 1024), pp = p % (PLANE_BYTES / 1024);
                const bool is_x = pp >= DY_BYTES / 1024;
                const int q = is_x ? pp - DY_BYTES / 1024 : pp;
                // byte offset of chunk block CPP*q of this operand tile inside its tile block
                const int64_t soff = tile32 * (is_x ? X_TILE : DY_TILE) + plane * (is_x ? X_PLANE : DY_PLANE) + (CPP * q) * 512 + half_off;
                const int voff = (CPP == 2 && (q & 1)) ? voff_odd : voff_even;
                // Issued from inline asm on purpose: hipcc orders every LDS read behind a
                // compiler-visible LDS-DMA with s_waitcnt vmcnt(0), which would drain the
                // whole prefetch ring at each tile.  M0 = LDS destination (wave-uniform),
                // saved/restored inside the statement; completion is tracked by the counted
                // s_waitcnt vmcnt(N) below (no other VMEM operation lives in the tile loop).
                const char* src = (is_x ? x_base : dy_base) + soff + (unsigned)voff;
                const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(dst + p * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" SP_WG_NT "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            }
        }
    };
    auto issue_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < PPW_HI; ++i) issue_piece(t, i);
    };

    // bf16: an MFMA operand fragment = 8 consecutive rows (k) of one column (lane&31), fetched with two
    // transposing LDS reads.  ds_read_b64_tr_b16: inside each 16-lane group the 16 x 4 b16 values loaded
    // from the lanes' own addresses are exchanged so that lane i receives element (i&3) of lanes 4j+(i>>2),
    // j = 0..3 (verified on hardware: tools/probes/tr_probe.hip).  Offsets of this lane: 16-lane group g covers
    // k-half g>>1 and feature half g&1; lane i of the group addresses row (i>>2), 8 bytes (i&1)
    // of chunk (g&1)*2 + ((i&3)>>1) of the 32-column block
    const int i16 = lane & 15, g = lane >> 4;
    const int c3 = (g & 1) * 2 + ((i16 & 3) >> 1);
    int roff[2];
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
        const int prow = ((((g >> 1) << 1 | q4) ^ c3) << 2) | (i16 >> 2);
        roff[q4] = c3 * CS + prow * 16 + (i16 & 1) * 8;
    }
    // fp32: lane l supplies element (row 2*kk + (l>>5), column l&31) of the 32-column block: chunk
    // (l&31)/4, float (l&31)%4; the row slot carries the same XOR swizzle as the DMA applied
    const int f_chunk = (lane & 31) >> 2, f_off = f_chunk * CS + (lane & 3) * 4, f_sw = (f_chunk & 3) << 2;
    auto frag = [&](const char* region, int kk, int blk) -> frag_t {
        if constexpr (FP32) {
            const int row = 2 * kk + h;
            return *(const float*)(region + blk * CPB * CS + f_off + ((row ^ f_sw) << 4));
        } else {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const char* base = region + blk * CPB * CS + kk * 256;           // k-step kk: rows 16*kk .. 16*kk+15 of the slot
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + roff[0]));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + roff[1]));
            s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, v);
        }
    };

    // Output blocks of this wave.  "n-owner" (NB a multiple of 8): the wave keeps one X fragment
    // (n-block = wave) and streams the MB dY fragments; "m-owner" (MB divides 8): it keeps one
    // dY fragment (m-block = wave % MB) and streams X fragments n = wave / MB + (8 / MB) * j.
    // Either way the tile loop is branch-free straight-line code: a block index past the end is
    // clamped (its product is computed and discarded) instead of branched around.
    constexpr bool N_OWNER = NB % 8 == 0;
    static_assert(N_OWNER || 8 % MB == 0, "job shape not covered by the wave assignment");
    constexpr int WPM = N_OWNER ? 1 : 8 / MB;                       // waves sharing an m-block
    constexpr int NJ = N_OWNER ? MB : (NB + WPM - 1) / WPM;         // blocks (accumulators) per wave
    constexpr int NBIAS = N_OWNER ? (MB + 7) / 8 : 1;               // bias m-blocks summed by this wave
    constexpr int NS = KSTEPS * NJ;                                 // streamed fragments per ring slot
    constexpr int PF = NS < 6 ? NS : 6;                             // fragments read ahead of their MFMA
    const int fix_blk = N_OWNER ? wave : wave % MB;
    auto str_blk = [&](int j) {
        if constexpr (N_OWNER) return j;
        else { const int nb = wave / MB + WPM * j; return nb < NB ? nb : NB - 1; }
    };

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum[NBIAS];
#pragma unroll
    for (int b = 0; b < NBIAS; ++b) bsum[b] = 0.f;

    // MFMAs of one ring slot: the transposing fragment reads, the products, the bias column sums
    // `mid(i)`: called behind MFMA i of the slot's NS (the ring refill under SP_WG_SPREAD)
    auto compute_tile = [&](const char* dy_t, auto&& mid) {
        const char* x_t = dy_t + DY_BYTES;
        const char* fix_t = N_OWNER ? x_t : dy_t;
        const char* str_t = N_OWNER ? dy_t : x_t;

        frag_t fx[KSTEPS][NPL], ring[PF][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) fx[kk][pl] = frag(fix_t + pl * PLANE_BYTES, kk, fix_blk);
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[i][pl] = frag(str_t + pl * PLANE_BYTES, i / NJ, str_blk(i % NJ));
        // bias gradient = column sums of dY: n-owner waves read "their" m-block once more,
        // m-owner waves already hold it
        frag_t bf[NBIAS][KSTEPS][NPL];
        if constexpr (N_OWNER) {
#pragma unroll
            for (int b = 0; b < NBIAS; ++b) {
                const int mb = wave + 8 * b < MB ? wave + 8 * b : MB - 1;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int kk = 0; kk < KSTEPS; ++kk) bf[b][kk][pl] = frag(dy_t + pl * PLANE_BYTES, kk, mb);
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) bf[0][kk][pl] = fx[kk][pl];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int kk = i / NJ, j = i % NJ;
            // A = dY fragment, B = X fragment; planes: [0] heads, [1] tails
            const frag_t* A_ = N_OWNER ? ring[i % PF] : fx[kk];
            const frag_t* B_ = N_OWNER ? fx[kk] : ring[i % PF];
            if constexpr (NPL == 2) {
                acc[j] = P::template mfma_part<0>(A_[1], B_[0], acc[j]);
                acc[j] = P::template mfma_part<0>(A_[0], B_[1], acc[j]);
            }
            acc[j] = P::template mfma_part<0>(A_[0], B_[0], acc[j]);
            if (i + PF < NS) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ring[i % PF][pl] = frag(str_t + pl * PLANE_BYTES, (i + PF) / NJ, str_blk((i + PF) % NJ));
            }
            mid(i);
            __builtin_amdgcn_sched_barrier(0);      // keep MFMA i, then the read for MFMA i + PF
        }
#pragma unroll
        for (int b = 0; b < NBIAS; ++b)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) bsum[b] += WOps<FP32 ? PREC_FP32 : PREC_BF16>::fsum(bf[b][kk][pl]);
    };
    if constexpr (!Q8) {
        for (int t = 0; t < DEPTH && t < ntiles; ++t) issue_tile(t);
        for (int t = 0; t < ntiles; ++t) {
            // wait for this wave's pieces of tile t: at most (tiles still in flight behind it)
            // x (pieces per tile) of its DMA operations may remain outstanding
            const int ahead = ntiles - 1 - t < DEPTH - 1 ? ntiles - 1 - t : DEPTH - 1;
            static_assert((DEPTH - 1) * PPW_HI <= 63, "vmcnt immediate");
            if (ahead == DEPTH - 1) {                                  // steady state
                if (wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PPW_HI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PPW_LO) : "memory");
            } else {                                                   // the last DEPTH-1 tiles of the range
                static_for<DEPTH - 1>([&](auto ac) {
                    constexpr int A = decltype(ac)::value;
                    if (ahead == A) {
                        if (wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A * PPW_HI) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A * PPW_LO) : "memory");
                    }
                });
            }
            __syncthreads();          // a bare s_barrier here: the compiler sees no VMEM in flight
            if constexpr (SP_WG_SPREAD) {
                // piece k of the refill behind MFMA k * NS / PPW_HI of this tile (i is a compile-time value once the loop is unrolled)
                const bool refill = t + DEPTH < ntiles;
                compute_tile(lds + (t % NBUF) * BUF_BYTES, [&](int i) {
#pragma unroll
                    for (int k = 0; k < PPW_HI; ++k)
                        if (i == k * NS / PPW_HI && refill) issue_piece(t + DEPTH, k);
                });
            } else {
                if (t + DEPTH < ntiles) issue_tile(t + DEPTH);
                compute_tile(lds + (t % NBUF) * BUF_BYTES, [](int) {});
            }
        }
    } else {
        // LDS: [bf16 image 0][bf16 image 1][QD ring slots of PQ 1 KiB 8-bit pieces][QD x 8 waves x {dY steps, X steps} (256 B each)]
        constexpr int PQ = MB + NB;                            // 1 KiB pieces per tile: dY blocks 0..MB-1, X blocks 0..NB-1
        constexpr int QSLOT = PQ * 1024, SSLOT = 8 * 512;
        constexpr int QD_FIT = (LDS_BYTES - 2 * BUF_BYTES) / (QSLOT + SSLOT), QD = QD_FIT < 4 ? QD_FIT : 4;     // tiles in flight
        static_assert(QD >= 2, "LDS budget of the 8-bit operand ring");
        constexpr int QP_HI = (PQ + 7) / 8, QP_LO = PQ / 8, QN_HI = PQ % 8;        // pieces per wave: waves < QN_HI take QP_HI
        constexpr int64_t DYQ_TILE = grad_tile_bytes(AREA_Q8), XQ_TILE = save_tile_bytes(AREA_Q8);
        const char* dyq = (const char*)a.grad + grad_buf_tile_off(AREA_Q8, jb.gbuf) + mb0 * 1024;       // (a 32-column block of the 8-bit areas = 1 KiB)
        const char* xq = (const char*)a.save + save_buf_tile_off(AREA_Q8, jb.sbuf) + (jb.xcol0 / 32) * 1024;
        const char* dys = (const char*)a.grad + grad_step_tile_off(jb.gbuf, 0);
        const char* xs = (const char*)a.save + save_step_tile_off(jb.sbuf, 0);
        char* ring8 = lds + 2 * BUF_BYTES;
        char* steps = ring8 + QD * QSLOT;
        const int row = lane & 31;
        auto dma = [&](const char* src, char* dst, auto widec) {
            const unsigned lds_dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst;
            unsigned keep;
            if constexpr (decltype(widec)::value)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" SP_WG_NT "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" SP_WG_NT "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        };
        // this wave's pieces of tile t and the two step rows it multiplies them back with (a wave converts what it fetched itself:
        // its own counted vmcnt is all the synchronisation the ring needs)
        // operation k of this wave for tile t: its QP_HI pieces, then the two step rows (QP_HI + 2 operations; the counted waits
        // below count operations, absent pieces of the waves >= QN_HI are not issued and not counted)
        auto issue_q8_op = [&](int t, int k) {
            const int64_t tile32 = (r_begin >> 5) + t;
            if (k < QP_HI) {
                char* slot = ring8 + (t % QD) * QSLOT;
                const int p = k * 8 + wave;                    // wave-uniform piece id
                if (p < PQ) {
                    const bool is_x = p >= MB;
                    const int C = is_x ? p - MB : p;
                    dma((is_x ? xq + tile32 * XQ_TILE : dyq + tile32 * DYQ_TILE) + C * 1024 + lane * 16, slot + p * 1024, std::true_type{});
                }
            } else {
                char* sw = steps + (t % QD) * SSLOT + wave * 512;
                if (k == QP_HI) dma(dys + tile32 * DYQ_TILE + lane * 4, sw, std::false_type{});             // [part 0][part 1] x 32 rows
                else dma(xs + tile32 * XQ_TILE + lane * 4, sw + 256, std::false_type{});
            }
        };
        auto issue_q8 = [&](int t) {
#pragma unroll
            for (int k = 0; k < QP_HI + 2; ++k) issue_q8_op(t, k);
        };
        auto convert_q8 = [&](int t, char* dst) {
            const char* slot = ring8 + (t % QD) * QSLOT;
            const char* sw = steps + (t % QD) * SSLOT + wave * 512;
#pragma unroll
            for (int i = 0; i < QP_HI; ++i) {
                const int p = i * 8 + wave;
                if (p < PQ) {
                    const bool is_x = p >= MB;
                    const int C = is_x ? p - MB : p;
                    const int part = (is_x ? jb.xcol0 / 32 + C : mb0 + C) >= 8 ? 1 : 0;       // the vector the block belongs to (layout.h AREA_Q8)
                    const u32x4 q = *(const u32x4*)(slot + p * 1024 + lane * 16);
                    const float step = *(const float*)(sw + (is_x ? 256 : 0) + part * 128 + row * 4), off = -128.0f * step;
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    u32x4 o[2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w = q[j];
                        const bf16x2_t lo = {(__bf16)fmaf((float)(w & 0xffu), step, off), (__bf16)fmaf((float)((w >> 8) & 0xffu), step, off)};
                        const bf16x2_t hi = {(__bf16)fmaf((float)((w >> 16) & 0xffu), step, off), (__bf16)fmaf((float)(w >> 24), step, off)};
                        o[j >> 1][2 * (j & 1)] = __builtin_bit_cast(unsigned, lo);
                        o[j >> 1][2 * (j & 1) + 1] = __builtin_bit_cast(unsigned, hi);
                    }
                    // slots [16 C, 16 C + 8) and [16 C + 8, 16 C + 16) of half h = bf16 chunk blocks 4 C + h and 4 C + 2 + h of the operand
                    char* region = dst + (is_x ? DY_BYTES : 0);
                    const int k0 = 4 * C + h, k1 = k0 + 2;
                    *(u32x4*)(region + k0 * CS + ((row ^ ((k0 & 3) << 2)) << 4)) = o[0];
                    *(u32x4*)(region + k1 * CS + ((row ^ ((k1 & 3) << 2)) << 4)) = o[1];
                }
            }
        };
        for (int t = 0; t < QD && t < ntiles; ++t) issue_q8(t);
        for (int t = 0; t < ntiles; ++t) {
            // this wave's operations of tile t have landed: at most (tiles in flight behind it) x (its operations per tile) remain
            const int ahead = ntiles - 1 - t < QD - 1 ? ntiles - 1 - t : QD - 1;
            static_for<QD>([&](auto ac) {
                constexpr int A = decltype(ac)::value;
                if (ahead == A) {
                    if (wave < QN_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A * (QP_HI + 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A * (QP_LO + 2)) : "memory");
                }
            });
            char* buf = lds + (t & 1) * BUF_BYTES;
            convert_q8(t, buf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // ring slot read, bf16 image written
            if constexpr (SP_WG_SPREAD) {
                asm volatile("s_barrier" ::: "memory");
                const bool refill = t + QD < ntiles;                        // (into the slot this wave has just converted out of)
                compute_tile(buf, [&](int i) {
#pragma unroll
                    for (int k = 0; k < QP_HI + 2; ++k)
                        if (i == k * NS / (QP_HI + 2) && refill) issue_q8_op(t + QD, k);
                });
            } else {
                if (t + QD < ntiles) issue_q8(t + QD);                          // (into the slot just consumed)
                asm volatile("s_barrier" ::: "memory");                         // bare barrier: the image of tile t is complete, that of t-1 free
                compute_tile(buf, [](int) {});
            }
        }
    }

    float* out = a.partial + (int64_t)blockIdx.x * WPARTIAL;
    float* mat = out + wjob_mat_off(job);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = N_OWNER ? j : fix_blk;
        const int nb_raw = N_OWNER ? wave : wave / MB + WPM * j;
        if (nb_raw < NB) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int po = 32 * (m + mb0) + (r & 3) + 8 * (r >> 2) + 4 * h;
                mat[(int64_t)po * N + nb_raw * 32 + (lane & 31)] = acc[j][r];
            }
        }
    }
    float* bo = out + wjob_bias_off(job);
#pragma unroll
    for (int b = 0; b < NBIAS; ++b) {
        const int m = N_OWNER ? wave + 8 * b : wave;       // m-owner: the first MB waves hold each m once
        const float sum = bsum[b] + __shfl_xor(bsum[b], 32);
        if (m < MB && lane < 32) bo[32 * (m + mb0) + lane] = sum;
    }
}

template <int PREC, bool Q8, int MB, int NB> SP_DEV void wgrad_dispatch(const WgradArgs& a, int job, char* lds) {
    if constexpr (Q8) {
        static_assert(PREC != PREC_FP32, "8-bit areas: bf16-operand modes");
        wgrad_job_dma<MB, NB, 1, 32, 2, true>(a, job, lds);
    } else if constexpr (PREC == PREC_BF16) {
        wgrad_job_dma<MB, NB>(a, job, lds);
    } else if constexpr (PREC == PREC_X3) {
#ifndef SP_WG_X3_ROWS
#define SP_WG_X3_ROWS 16
#endif
        if constexpr (nplanes_of(PREC_X3) == 2) wgrad_job_dma<MB, NB, 2, SP_WG_X3_ROWS>(a, job, lds);
        else wgrad_job_dma<MB, NB>(a, job, lds);          // head planes only: the bf16 kernel on them
    } else {
        wgrad_job_dma<MB, NB, 1, 16, 4>(a, job, lds);     // fp32: LDS-DMA ring of 16-row slots (8.9 -> 7.3 ms vs register staging)
    }
}

// Experiment (round 6; -DSP_WG_Q8_HALVES=1, off): the 8-bit-operand jobs whose dY operand is 8 blocks wide -- layers 0-3, 5, 6:
// two thirds of the launch's bytes -- as HALF jobs, two resident workgroups per CU (80 KiB of LDS, 128 VGPRs each, no spills: the geometry
// round 5 named as "what would overlap the chain").  Each half fetches its four dY blocks and ALL of X: 1.5 x the bytes of those jobs, for the
// chance that one workgroup's chain (land -> convert -> barrier -> fragments -> MFMAs) overlaps the other's.  Same partial blocks, same
// summation order per output element: bit-identical (tests/test_q8_saves_gpu.py with the flag on: 9 passed).
// MEASURED AND NOT ADOPTED (same box, three repetitions, 786 432 rows, profiles/r06_wgrad_q8_halves.log): weight-gradient launch bf16x3+q8
// 1.402-1.406 -> 1.604-1.610 ms, bf16+q8 1.424-1.429 -> 1.622-1.628 ms; config-1 step 6.66 -> 7.03 ms (615 k -> 583 k rays/s).  The split
// moves 1.28 x the bytes (3.73 -> 4.79 GB) in 1.15 x the time: a second resident workgroup buys ~12 % per byte, the shared operand fetched
// twice costs 28 %.  With whole jobs two workgroups do not fit (64 output blocks = 128 accumulator registers per wave at 8 waves, 256 at 4).
#ifndef SP_WG_Q8_HALVES
#define SP_WG_Q8_HALVES 0
#endif
enum { WGRAD_H_LDS_BYTES = 80 * 1024 };
enum : unsigned { WGRAD_H_JOBS = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 3) | (1u << 5) | (1u << 6) };
__global__ void __launch_bounds__(WG_THREADS, 4) wgrad_q8h_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[WGRAD_H_LDS_BYTES];
    const int job = blockIdx.y, mb0 = 4 * (int)blockIdx.z;
    if (!((WGRAD_H_JOBS >> job) & 1u)) return;
    if (job == 0) wgrad_job_dma<4, 2, 1, 32, 2, true, WGRAD_H_LDS_BYTES>(a, job, lds, mb0);
    else wgrad_job_dma<4, 8, 1, 32, 2, true, WGRAD_H_LDS_BYTES>(a, job, lds, mb0);
}

// job_mask: the jobs this launch computes (bit j = job j); the others are another launch's (wgrad_q8h_kernel)
template <int PREC, bool Q8 = false>
__global__ void __launch_bounds__(WG_THREADS) wgrad_kernel(WgradArgs a, unsigned job_mask) {
    // the whole 160 KiB LDS of the CU, declared statically: gfx950 launches 163 840 B of static LDS
    // without the per-function opt-in dynamic LDS above 64 KiB would need (no host-side state)
    __shared__ __attribute__((aligned(16))) char lds[WGRAD_LDS_BYTES];
    const int job = blockIdx.y;
    if (!((job_mask >> job) & 1u)) return;
    switch (job) {
        case 0: wgrad_dispatch<PREC, Q8, 8, 2>(a, job, lds); break;
        case 4: wgrad_dispatch<PREC, Q8, 8, 10>(a, job, lds); break;
        case 7: wgrad_dispatch<PREC, Q8, 9, 8>(a, job, lds); break;
        case 8: wgrad_dispatch<PREC, Q8, 4, 9>(a, job, lds); break;
        case 9: wgrad_dispatch<PREC, Q8, 1, 4>(a, job, lds); break;
        default: wgrad_dispatch<PREC, Q8, 8, 8>(a, job, lds); break;
    }
}

// out[p] = sum_s partial[s][wsrc[p]]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, const int32_t* __restrict__ wsrc,
                                    float* __restrict__ out, int accumulate) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N_PARAMS) return;
    constexpr int64_t WPARTIAL = wpartial_floats();
    const int64_t src = wsrc[p];
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * WPARTIAL + src];
    out[p] = accumulate ? out[p] + s : s;
}

// the split-K partial products of `nsplit` row ranges (a.partial = the first of their partial blocks)
int launch_wgrad_partials(int prec, bool q8, const WgradArgs& a, int nsplit, hipStream_t s) {
    if (a.rows <= 0 || nsplit <= 0) return 1;
    dim3 grid(nsplit, N_WJOBS), block(WG_THREADS);
    const unsigned all = ~0u;
    if (q8) {            // one kernel for both bf16-operand modes: the areas are the same
        if (prec != PREC_BF16 && prec != PREC_X3) return 1;
        if (SP_WG_Q8_HALVES) {
            hipLaunchKernelGGL(wgrad_q8h_kernel, dim3(nsplit, N_WJOBS, 2), block, 0, s, a);
            hipLaunchKernelGGL((wgrad_kernel<PREC_BF16, true>), grid, block, 0, s, a, all & ~(unsigned)WGRAD_H_JOBS);
        } else hipLaunchKernelGGL((wgrad_kernel<PREC_BF16, true>), grid, block, 0, s, a, all);
    } else if (prec == PREC_BF16) hipLaunchKernelGGL(wgrad_kernel<PREC_BF16>, grid, block, 0, s, a, all);
    else if (prec == PREC_X3) hipLaunchKernelGGL(wgrad_kernel<PREC_X3>, grid, block, 0, s, a, all);
    else if (prec == PREC_FP32) hipLaunchKernelGGL(wgrad_kernel<PREC_FP32>, grid, block, 0, s, a, all);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
// grad_out (+)= sum over `nsplit` partial blocks, un-permuted into nn.Linear order
int launch_wgrad_reduce(const float* partial, int nsplit, const int32_t* wsrc, float* grad_out, hipStream_t s, bool accumulate) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((N_PARAMS + 255) / 256), dim3(256), 0, s, partial, nsplit, wsrc, grad_out, accumulate ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int launch_wgrad(int prec, bool q8, const WgradArgs& a, int nsplit, const int32_t* wsrc, float* grad_out, hipStream_t s, bool accumulate) {
    const int rc = launch_wgrad_partials(prec, q8, a, nsplit, s);
    return rc ? rc : launch_wgrad_reduce(a.partial, nsplit, wsrc, grad_out, s, accumulate);
}

}  // namespace sparf
