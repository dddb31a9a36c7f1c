// Host-side builder of the static gather tables (one set per precision).
//
// Every permutation the kernels rely on (weight fragment order for the forward and the
// transposed/backward MFMA streams, packed accumulator-initial biases, and the
// "un-permute" map from wgrad partial sums back to nn.Linear [out][in] order) is derived
// here from layout.h / streams.h and handed to the device as flat int32 index arrays.
// No device code in this file: it runs (and is unit-tested) without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "streams.h"

namespace sparf {

// index into the flat parameter space of W_l[out_row][col], or -1
static inline int32_t widx(int l, int out_row, int col) {
    if (out_row < 0 || col < 0) return -1;
    return (int32_t)(param_w_off(l) + (int64_t)out_row * layer_in(l) + col);
}

// weight column addressed by slot (q,h) of input segment s of layer l, or -1
static inline int in_col(int l, int s, int q, int h) {
    int kind = layer_seg_kind(l, s);
    if (2 * q >= vk_width(kind)) return -1;
    int f = vk_feat(kind, q, h);
    if (f < 0) return -1;
    int ref_w = kind == VK_X0 ? X0_DIM : kind == VK_VIEW ? V_DIM : vk_width(kind);
    if (f >= ref_w) return -1;
    return layer_seg_coloff(l, s) + f;
}

static void fill_chunk(int prec, bool bwd, const Chunk& c, int32_t* out) {
    const int KJ = kj_of(prec);
    const int frag_elems = 64 * KJ;
    const int n = chunk_bytes(prec, c) / abytes_of(prec);
    for (int i = 0; i < n; ++i) out[i] = -1;
    for (int ks = 0; ks < c.nks; ++ks)
        for (int m = 0; m < c.nmb; ++m) {
            int32_t* f = out + (int64_t)(ks * c.nmb + m) * frag_elems;
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < KJ; ++jj) {
                    int crow_m = 32 * (c.mb0 + m) + (lane & 31);   // MFMA M index
                    int qk = (c.ks0 + ks) * KJ + jj, hk = lane >> 5;  // MFMA K slot
                    int32_t v;
                    if (!bwd) {
                        // A = W: M = output C-row, K = input slot
                        v = widx(c.layer, out_row_of_crow(c.layer, crow_m), in_col(c.layer, c.seg, qk, hk));
                        // bf16x3: the raw-coordinate columns of the encoded point leave the MFMA stream (streams.h xyz_pk)
                        if (xyz_exact(prec) && layer_seg_kind(c.layer, c.seg) == VK_X0 && 2 * qk < X0_W && x0_feat(qk, hk) >= 0 && x0_feat(qk, hk) < 3) v = -1;
                    } else {
                        // A = W^T: M = input C-row of segment, K = output slot
                        int out_row = out_row_of_crow(c.layer, crow_of(qk, hk));
                        if (crow_of(qk, hk) >= 32 * layer_out_mb(c.layer)) out_row = -1;
                        v = widx(c.layer, out_row, in_col(c.layer, c.seg, q_of_crow(crow_m), h_of_crow(crow_m)));
                    }
                    f[lane * KJ + jj] = v;
                }
        }
}

// (kind, local pos) of absolute column `col` of saved buffer `sbuf`
static inline void save_col_kind(int sbuf, int col, int* kind, int* local) {
    if (sbuf == SB_XS) { *kind = col < 256 ? VK_HID256 : VK_X0; *local = col < 256 ? col : col - 256; }
    else if (sbuf == SB_FV) { *kind = col < 256 ? VK_HID256 : VK_VIEW; *local = col < 256 ? col : col - 256; }
    else if (sbuf == SB_G) { *kind = VK_HID128; *local = col; }
    else { *kind = VK_HID256; *local = col; }
}

int build_tables(int prec, int32_t* out) {
    if (prec < 0 || prec >= N_PREC) return 1;
    const int CH = ch_of(prec);
    // forward / backward weight streams
    for (int id = 0; id < fwd_nchunks(prec); ++id)
        fill_chunk(prec, false, fwd_chunk(prec, id), out + tbl_fwd_off(prec) + fwd_chunk_off(prec, id) / abytes_of(prec));
    for (int id = 0; id < bwd_nchunks(prec); ++id)
        fill_chunk(prec, true, bwd_chunk(prec, id), out + tbl_bwd_off(prec) + bwd_chunk_off(prec, id) / abytes_of(prec));
    // packed biases [layer][mb][h][r]
    int32_t* bp = out + tbl_bias_off(prec);
    for (int l = 0; l < N_LAYERS; ++l)
        for (int mb = 0; mb < layer_out_mb(l); ++mb)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    int row = out_row_of_crow(l, crow_of(16 * mb + r, h));
                    bp[bias_pk_off(l) + mb * 32 + h * 16 + r] = row < 0 ? -1 : (int32_t)(param_b_off(l) + row);
                }
    // raw-coordinate columns of layers 0 / 4 in the packed-bias order (streams.h xyz_pk; read by bf16x3 only)
    for (int which = 0; which < 2; ++which)
        for (int coord = 0; coord < 3; ++coord)
            for (int mb = 0; mb < 8; ++mb)
                for (int h = 0; h < 2; ++h)
                    for (int r = 0; r < 16; ++r) {
                        const int l = which == 0 ? 0 : 4;
                        const int row = out_row_of_crow(l, crow_of(16 * mb + r, h));
                        bp[xyz_pk_off(which, coord) + mb * 32 + h * 16 + r] = widx(l, row, (which == 0 ? 0 : 256) + coord);
                    }
    // wgrad un-permute: parameter index -> offset inside one split's partial block
    int32_t* ws = out + tbl_wsrc_off(prec);
    for (int i = 0; i < N_PARAMS; ++i) ws[i] = -1;
    bool bias_done[N_LAYERS] = {false};
    for (int j = 0; j < N_WJOBS; ++j) {
        const WJob jb = wjob(j);
        const int M = 32 * jb.mb, N = 32 * jb.nb, l = jb.layer;
        for (int po = 0; po < M; ++po) {
            int crow_o = crow_of(q_of_pos(po, CH), h_of_pos(po, CH));
            int out_row = out_row_of_crow(l, crow_o);
            if (out_row < 0) continue;
            if (!bias_done[l]) ws[param_b_off(l) + out_row] = (int32_t)(wjob_bias_off(j) + po);
            for (int pi = 0; pi < N; ++pi) {
                int kind, local;
                save_col_kind(jb.sbuf, jb.xcol0 + pi, &kind, &local);
                int s = -1;
                for (int t = 0; t < layer_nseg(l); ++t)
                    if (layer_seg_kind(l, t) == kind) s = t;
                if (s < 0) return 2;
                int col = in_col(l, s, q_of_pos(local, CH), h_of_pos(local, CH));
                if (col < 0) continue;
                ws[param_w_off(l) + (int64_t)out_row * layer_in(l) + col] = (int32_t)(wjob_mat_off(j) + (int64_t)po * N + pi);
            }
        }
        bias_done[l] = true;
    }
    for (int i = 0; i < N_PARAMS; ++i)
        if (ws[i] < 0) return 3;            // every parameter must receive a gradient
    return 0;
}

}  // namespace sparf
