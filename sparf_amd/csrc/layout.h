// Layout algebra shared by host table builders and device kernels.
//
// The MLP is evaluated "transposed": for a wave's 32 sample rows the MFMA computes
//     H_l^T [features x rows] = W_l [out x in] * H_{l-1}^T [in x rows]
// with A = weights (LDS), B = activations (registers), D = activations of the next
// layer.  For v_mfma_f32_32x32x{16_bf16,2_f32} lane l, accumulator register r of an
// m-block holds D[row i][col n] with n = l&31 and i = (r&3) + 8*(r>>2) + 4*(l>>5),
// and the B operand of lane l carries column n = l&31 again: a wave's D registers can
// be re-used as the next layer's B operand WITHOUT any cross-lane movement, provided
// the weights are packed with the matching permutation of the contraction index.
// Everything below is that permutation, written once.
//
// Vocabulary
//   crow : index inside a layer's output "C-row space" (0..32*MB-1), the MFMA M index
//   h    : lane half (l>>5)
//   q    : per-lane-half register slot, q = 16*mblock + r  (so a 256-wide vector has
//          q in 0..127 on each half)
//   pos  : column inside a saved [rows][cols] activation buffer in HBM
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ inline
#else
#define SP_HD inline
#endif

namespace sparf {

// Precision modes.  PREC_X3 ("bf16x3"): every fp32 operand is split into a bf16 head and a
// bf16 tail (x = hi + lo, 16 mantissa bits) and a product is three bf16 MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulate): bf16-MFMA rate / 3 at ~2e-5 relative error.  It
// shares the bf16 register / fragment layout (KJ = 8); its saved activations are two bf16
// planes (all heads, then all tails), its weight fragments are [1 KiB heads][1 KiB tails].
enum { PREC_BF16 = 0, PREC_FP32 = 1, PREC_X3 = 2, N_PREC = 3 };

// ---- C-row <-> (q, h) ---------------------------------------------------------------
SP_HD constexpr int crow_of(int q, int h) { return 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h; }
SP_HD constexpr int q_of_crow(int c) { return 16 * (c >> 5) + (c & 3) + 4 * ((c & 31) >> 3); }
SP_HD constexpr int h_of_crow(int c) { return (c >> 2) & 1; }

// ---- per-precision constants ----------------------------------------------------------
// KJ  : contraction elements one lane supplies per MFMA (8 bf16 / 1 f32)
// CH  : elements per 16-byte chunk of a saved activation row
template <int PREC> struct PrecInfo;
template <> struct PrecInfo<PREC_BF16> { enum { KJ = 8, CH = 8, ABYTES = 2, FRAG_BYTES = 1024 }; };
template <> struct PrecInfo<PREC_FP32> { enum { KJ = 1, CH = 4, ABYTES = 4, FRAG_BYTES = 256 }; };
template <> struct PrecInfo<PREC_X3> { enum { KJ = 8, CH = 8, ABYTES = 4, FRAG_BYTES = 2048 }; };

SP_HD constexpr int kj_of(int prec) { return prec == PREC_FP32 ? 1 : 8; }
SP_HD constexpr int ch_of(int prec) { return prec == PREC_FP32 ? 4 : 8; }
// bytes per logical element of a weight stream (x3: head + tail)
SP_HD constexpr int abytes_of(int prec) { return prec == PREC_BF16 ? 2 : 4; }
// Saved activations / gradients (the wgrad operands).  bf16x3 keeps only the HEAD plane by
// default: the weight gradient is a sum over rows of products of already-computed values, its
// operands' bf16 rounding is unbiased and averages out (relative error ~ 2^-8 / sqrt(rows), far
// below the ReLU-flip noise of the mode), whereas the forward and data-gradient chains, which
// propagate errors, keep all three partial products.  -DSP_X3_SAVE_PLANES=2 stores the tails too.
#ifndef SP_X3_SAVE_PLANES
#define SP_X3_SAVE_PLANES 1
#endif
// AREA FORMATS.  The save / gradient areas of a pass are laid out per "area format" af: a precision id (planes of that
// precision's element type, as above) or AREA_Q8 -- the 8-bit format of the bf16-operand modes (sparf_hip.h SPARF_SAVE_Q8):
// every saved vector of a row as signed 8-bit integers on a LINEAR grid with one fp32 step per row and vector,
//     x ~ (u - 128) * step,   step = max |x| over the vector / 127,   u in 1..255,
// half the bytes of the bf16 planes for 1.75x their backward error (tests/tools/save_precision_study.py: a sum over rows
// wants a linear grid with a per-row scale, not an 8-bit float).  Every function below that takes `prec` to size or address an
// area takes an area format.
enum { AREA_Q8 = 3 };
SP_HD constexpr int area_format(int prec, bool q8) { return q8 ? (int)AREA_Q8 : prec; }
SP_HD constexpr int nplanes_of(int prec) { return prec == PREC_X3 ? SP_X3_SAVE_PLANES : 1; }
SP_HD constexpr int plane_ebytes_of(int prec) { return prec == PREC_FP32 ? 4 : prec == AREA_Q8 ? 1 : 2; }
// bytes per logical element of a saved row (all planes)
SP_HD constexpr int save_abytes_of(int prec) { return nplanes_of(prec) * plane_ebytes_of(prec); }
SP_HD constexpr int frag_bytes_of(int prec) { return prec == PREC_BF16 ? 1024 : prec == PREC_FP32 ? 256 : 2048; }

// column of (q,h) inside a saved activation row: lanes write 16-byte chunks, the two
// halves of a row interleave chunk-wise.
SP_HD constexpr int pos_of(int q, int h, int ch) { return (q / ch) * (2 * ch) + h * ch + (q % ch); }
SP_HD constexpr int q_of_pos(int pos, int ch) { return (pos / (2 * ch)) * ch + pos % ch; }
SP_HD constexpr int h_of_pos(int pos, int ch) { return (pos / ch) & 1; }

// ---- architecture (the shipped SPARF network) ---------------------
// mlp_feat: 63->256, 256->256 x3, 319->256 (skip: [h, x0]), 256->256 x2, 256->257
// mlp_rgb : 283->128 ([feat, view27]), 128->3
enum { N_LAYERS = 10, L3D = 10, LVIEW = 4, X0_DIM = 63, V_DIM = 27, HID = 256, RGB_HID = 128 };
enum { X0_W = 64, V_W = 32 };   // padded vector widths (multiples of 32 C-rows)

SP_HD constexpr int layer_out(int l) { return l < 7 ? 256 : l == 7 ? 257 : l == 8 ? 128 : 3; }
SP_HD constexpr int layer_in(int l) { return l == 0 ? 63 : l == 4 ? 319 : l < 8 ? 256 : l == 8 ? 283 : 128; }
// offset of layer l's weight / bias in the flat parameter space (W0,b0,W1,b1,...)
SP_HD constexpr int64_t param_w_off(int l) {
    int64_t o = 0;
    for (int i = 0; i < l; ++i) o += (int64_t)layer_out(i) * layer_in(i) + layer_out(i);
    return o;
}
SP_HD constexpr int64_t param_b_off(int l) { return param_w_off(l) + (int64_t)layer_out(l) * layer_in(l); }
enum { N_PARAMS = 530052 };   // per network, without the scalar `progress`

// Output C-row space of each layer, in m-blocks, and C-row -> weight row (or -1).
//   l<7 : 8 blocks, identity.   l==7: 9 blocks, crow<256 -> W7 row crow+1 (feature),
//   crow==256 -> W7 row 0 (raw sigma).   l==8: 4 blocks.   l==9: 1 block, rows 0..2.
SP_HD constexpr int layer_out_mb(int l) { return l < 7 ? 8 : l == 7 ? 9 : l == 8 ? 4 : 1; }
SP_HD constexpr int out_row_of_crow(int l, int crow) {
    return l < 7 ? crow
         : l == 7 ? (crow < 256 ? crow + 1 : crow == 256 ? 0 : -1)
         : l == 8 ? crow
         : (crow < 3 ? crow : -1);
}

// Input vectors.  A layer input is one or two segments; each segment is a vector kind.
enum VecKind { VK_HID256 = 0, VK_X0 = 1, VK_VIEW = 2, VK_HID128 = 3 };
SP_HD constexpr int vk_width(int vk) { return vk == VK_HID256 ? 256 : vk == VK_X0 ? 64 : vk == VK_VIEW ? 32 : 128; }

// encoded point x0 (reference order: [p(3), per coord: 10 sin, 10 cos]); each lane half
// evaluates 15 of the 30 (coord,freq) arguments with one sincos each.
SP_HD constexpr int x0_feat(int q, int h) {
    if (q < 30) {
        int a = 15 * h + (q >> 1);
        return 3 + (a / 10) * 20 + (q & 1) * 10 + (a % 10);
    }
    if (q == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;          // q == 31
}
// encoded view direction (reference order: [d(3), per coord: 4 sin, 4 cos])
SP_HD constexpr int view_feat(int q, int h) {
    if (q < 12) {
        int a = 6 * h + (q >> 1);
        return 3 + (a / 4) * 8 + (q & 1) * 4 + (a % 4);
    }
    if (q == 12) return h == 0 ? 0 : 2;
    if (q == 13) return h == 0 ? 1 : -1;
    return -1;
}
// feature index (inside the segment's reference vector) carried by slot (q,h), or -1
SP_HD constexpr int vk_feat(int vk, int q, int h) {
    return vk == VK_X0 ? x0_feat(q, h) : vk == VK_VIEW ? view_feat(q, h) : crow_of(q, h);
}

// segments of layer l's input: kind and column offset in the weight matrix
SP_HD constexpr int layer_nseg(int l) { return (l == 4 || l == 8) ? 2 : 1; }
SP_HD constexpr int layer_seg_kind(int l, int s) {
    return l == 0 ? VK_X0 : l == 4 ? (s == 0 ? VK_HID256 : VK_X0) : l == 8 ? (s == 0 ? VK_HID256 : VK_VIEW)
         : l == 9 ? VK_HID128 : VK_HID256;
}
SP_HD constexpr int layer_seg_coloff(int l, int s) { return s == 0 ? 0 : 256; }

// ---- saved activation / gradient buffers (columns per row) ---------------------------
// forward saves (inputs of every layer, for relu masks and wgrad):
//   XS [320] = [h3 | x0], H0,H1,H2,H4,H5,H6 [256], FV [288] = [feat | view], G [128]
// backward writes (pre-activation gradients, wgrad A operands):
//   DY0..DY6 [256], DY7 [288] (block 8 = d raw_sigma at q=128,h=0), DG [128], DZ [32]
enum { SAVE_COLS = 320 + 6 * 256 + 288 + 128, GRAD_COLS = 7 * 256 + 288 + 128 + 32 };
enum SaveBuf { SB_XS = 0, SB_H0, SB_H1, SB_H2, SB_H4, SB_H5, SB_H6, SB_FV, SB_G, SB_COUNT };
SP_HD constexpr int save_cols(int b) { return b == SB_XS ? 320 : b == SB_FV ? 288 : b == SB_G ? 128 : 256; }
SP_HD constexpr int64_t save_coloff(int b) {
    int64_t o = 0;
    for (int i = 0; i < b; ++i) o += save_cols(i);
    return o;
}
enum GradBuf { GB_DY0 = 0, GB_DY1, GB_DY2, GB_DY3, GB_DY4, GB_DY5, GB_DY6, GB_DY7, GB_DG, GB_DZ, GB_COUNT };
SP_HD constexpr int grad_cols(int b) { return b < GB_DY7 ? 256 : b == GB_DY7 ? 288 : b == GB_DG ? 128 : 32; }
SP_HD constexpr int64_t grad_coloff(int b) {
    int64_t o = 0;
    for (int i = 0; i < b; ++i) o += grad_cols(i);
    return o;
}
// AREAS.  The save area (forward -> dgrad / wgrad) and the gradient area (dgrad -> wgrad) are
// TILE-BLOCK-major: everything a 32-row tile owns is one contiguous block,
//     save area : [tile32][plane][buffer b][16-byte chunk c][row&31][CH elements] ... then SB_COUNT mask KiB
//     grad area : [tile32][plane][buffer b][16-byte chunk c][row&31][CH elements]
// i.e. inside a buffer exactly the register image of a wave (32 rows x CH-element chunks): every
// 16-byte store / load instruction of the fused kernels covers 1 KiB of contiguous memory, and a
// 32-row x C-column operand tile of the wgrad kernel is one contiguous block.  A wave addresses its
// tile through ONE buffer descriptor (base = area + tile32 * tile bytes, 64-bit scalar arithmetic)
// with compile-time offsets inside the block, so a launch has no 2^31-byte limit and the kernels keep
// four descriptor SGPRs instead of one descriptor per saved buffer.
// Rows are padded to the largest workgroup tile (8 waves x 32 rows), so that every wave of the
// fused kernels stores whole 32-row tiles unconditionally.
SP_HD constexpr int64_t rows_padded(int64_t rows) { return (rows + 255) & ~(int64_t)255; }
SP_HD constexpr int64_t ntiles32(int64_t rows) { return rows_padded(rows) / 32; }

// ReLU masks.  Next to the saved buffers of a tile the forward kernel stores which elements of
// every layer output are > 0, one bit per element, as a FIFO of bits per lane: lane (n, h) pushes
// its elements in register order (m-block ascending, register r ascending) into the LOW end of a
// 32-bit word (v_addc_co_u32 word, word, word, carry = [x > 0]), two m-blocks per word, so element
// e = 16 * (mb & 1) + r of word mb / 2 ends at bit 31 - e; the dgrad kernel pops them from the HIGH
// end in the same order (v_add_co_u32 word, word, word -> carry = the element's bit, consumed by
// v_cndmask).  A lane's four words of a buffer are contiguous: [lane 0..63][pair p 0..3][4 B] =
// 1 KiB per tile and buffer (the 128-wide G uses 2 pairs): one 16-byte store per lane and layer in
// the forward, one 16-byte load in the dgrad kernel, which reads these 32 B/row/layer instead of
// the 512 B/row/layer activations.
enum { MASK_TILE_BYTES = 1024 };
// AREA_Q8: a buffer of a tile is [32-column block C][lane half h][row&31][16 bytes = slots q in [16 C, 16 C + 16) of that half]
// (a lane of the fused kernels stores 16 bytes = two of its bf16x8 k-step chunks; 1 KiB per store instruction as before, half as
// many of them), followed -- after the mask words -- by the steps: [buffer][part 0 / 1][row&31] fp32, part 1 = the columns from
// 256 on (x0 behind h3 in XS, the view encoding behind feat in FV, the raw-density slot of DY7), which are vectors of their own.
enum { Q8_STEP_BYTES = 2 * 32 * 4 };        // per buffer and tile
SP_HD constexpr int64_t save_plane_tile_bytes(int af) { return (int64_t)SAVE_COLS * 32 * plane_ebytes_of(af); }
SP_HD constexpr int64_t grad_plane_tile_bytes(int af) { return (int64_t)GRAD_COLS * 32 * plane_ebytes_of(af); }
SP_HD constexpr int64_t save_tile_bytes(int af) {
    return nplanes_of(af) * save_plane_tile_bytes(af) + SB_COUNT * MASK_TILE_BYTES + (af == AREA_Q8 ? SB_COUNT * Q8_STEP_BYTES : 0);
}
SP_HD constexpr int64_t grad_tile_bytes(int af) { return nplanes_of(af) * grad_plane_tile_bytes(af) + (af == AREA_Q8 ? GB_COUNT * Q8_STEP_BYTES : 0); }
// byte offsets inside a tile block: buffer b (head plane), its mask KiB, its steps (AREA_Q8)
SP_HD constexpr int save_buf_tile_off(int af, int b) { return (int)(save_coloff(b) * 32 * plane_ebytes_of(af)); }
SP_HD constexpr int grad_buf_tile_off(int af, int b) { return (int)(grad_coloff(b) * 32 * plane_ebytes_of(af)); }
SP_HD constexpr int save_mask_tile_off(int af, int b) { return (int)(nplanes_of(af) * save_plane_tile_bytes(af)) + b * MASK_TILE_BYTES; }
SP_HD constexpr int save_step_tile_off(int b, int part) { return save_mask_tile_off(AREA_Q8, SB_COUNT) + b * Q8_STEP_BYTES + part * 128; }
SP_HD constexpr int grad_step_tile_off(int b, int part) { return (int)grad_plane_tile_bytes(AREA_Q8) + b * Q8_STEP_BYTES + part * 128; }
SP_HD constexpr int64_t save_area_bytes(int af, int64_t rows) { return ntiles32(rows) * save_tile_bytes(af); }
SP_HD constexpr int64_t grad_area_bytes(int af, int64_t rows) { return ntiles32(rows) * grad_tile_bytes(af); }

// ---- wgrad jobs: dW[pos_out][pos_in] = sum_rows DY[row][pos_out] * X[row][pos_in] -----
// job : layer, DY buffer (MB m-blocks), X view (buffer, first column, NB n-blocks),
//       weight column offset of that input segment
struct WJob { int layer, gbuf, mb, sbuf, xcol0, nb; };
enum { N_WJOBS = 10 };
SP_HD constexpr WJob wjob(int j) {
    return j == 0 ? WJob{0, GB_DY0, 8, SB_XS, 256, 2}
         : j == 1 ? WJob{1, GB_DY1, 8, SB_H0, 0, 8}
         : j == 2 ? WJob{2, GB_DY2, 8, SB_H1, 0, 8}
         : j == 3 ? WJob{3, GB_DY3, 8, SB_H2, 0, 8}
         : j == 4 ? WJob{4, GB_DY4, 8, SB_XS, 0, 10}    // skip layer: [h3 | x0] in one 320-wide view (DY4 read once)
         : j == 5 ? WJob{5, GB_DY5, 8, SB_H4, 0, 8}
         : j == 6 ? WJob{6, GB_DY6, 8, SB_H5, 0, 8}
         : j == 7 ? WJob{7, GB_DY7, 9, SB_H6, 0, 8}
         : j == 8 ? WJob{8, GB_DG, 4, SB_FV, 0, 9}      // [feat | view] in one 288-wide view
         : WJob{9, GB_DZ, 1, SB_G, 0, 4};
}
// partial-sum block of one split: per job an [32*mb][32*nb] fp32 matrix, then per job a
// [32*mb] bias-gradient vector (only taken from the first job of each layer).
SP_HD constexpr int64_t wjob_mat_off(int j) {
    int64_t o = 0;
    for (int i = 0; i < j; ++i) o += (int64_t)1024 * wjob(i).mb * wjob(i).nb;
    return o;
}
SP_HD constexpr int64_t wjob_bias_off(int j) {
    int64_t o = wjob_mat_off(N_WJOBS);
    for (int i = 0; i < j; ++i) o += 32 * wjob(i).mb;
    return o;
}
SP_HD constexpr int64_t wpartial_floats() { return wjob_bias_off(N_WJOBS); }

}  // namespace sparf
