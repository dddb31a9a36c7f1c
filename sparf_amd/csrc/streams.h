// Weight-fragment streams: the order in which the fused MLP kernels consume A-operand
// fragments, cut into LDS-sized chunks.  Host (table builder, pack.hip) and device
// (mlp_fwd.hip / mlp_bwd.hip) both derive chunk ids and sizes from the constexpr functions
// here, so they cannot disagree.
//
// A fragment is one MFMA A operand: 32 C-rows x (2*KJ) contraction slots, stored
// lane-major (lane l's KJ elements contiguous) -> 1 KiB (bf16) / 256 B (f32).
// A chunk holds fragments in [k-step][m-block] order for `nmb` m-blocks and `nks`
// k-steps and is at most CHUNK_MAX_BYTES (32 KiB); the kernels double-buffer chunks in LDS.
#pragma once
#include "layout.h"

namespace sparf {

enum { CHUNK_MAX_BYTES = 32768, X0_STASH_BYTES = 32768 };

SP_HD constexpr int group_g(int prec) { return 2; }   // m-blocks per accumulator group
// waves per workgroup of the fused MLP kernels (each wave owns 32 sample rows): bf16 runs
// 2 waves per SIMD inside the 256-VGPR budget, fp32 needs the whole 512-register file.
SP_HD constexpr int nwaves_of(int prec) { return prec == PREC_BF16 ? 8 : 4; }      // x3: head + tail registers, as fp32
SP_HD constexpr int seg_nks(int prec, int kind) { return vk_width(kind) / 2 / kj_of(prec); }
SP_HD constexpr int round_up_1k(int b) { return (b + 1023) & ~1023; }
SP_HD constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
// most k-steps one chunk can hold (G m-blocks per k-step)
SP_HD constexpr int nks_max(int prec) { return CHUNK_MAX_BYTES / (group_g(prec) * frag_bytes_of(prec)); }

struct Chunk { int layer, seg, mb0, nmb, ks0, nks; };
SP_HD constexpr int chunk_bytes(int prec, const Chunk& c) { return round_up_1k(c.nks * c.nmb * frag_bytes_of(prec)); }

// ------------------------------------------------------------------ forward stream
// for l in 0..9: for g in groups(out m-blocks): for s in input segments: for kp in k-parts.
SP_HD constexpr int fwd_ngroups(int prec, int l) { return cdiv(layer_out_mb(l), group_g(prec)); }
SP_HD constexpr int fwd_seg_nparts(int prec, int l, int s) { return cdiv(seg_nks(prec, layer_seg_kind(l, s)), nks_max(prec)); }
SP_HD constexpr int fwd_chunks_per_group(int prec, int l) {
    int n = 0;
    for (int s = 0; s < layer_nseg(l); ++s) n += fwd_seg_nparts(prec, l, s);
    return n;
}
SP_HD constexpr int fwd_chunk_id(int prec, int l, int g, int s, int kp) {
    int id = 0;
    for (int i = 0; i < l; ++i) id += fwd_ngroups(prec, i) * fwd_chunks_per_group(prec, i);
    id += g * fwd_chunks_per_group(prec, l);
    for (int t = 0; t < s; ++t) id += fwd_seg_nparts(prec, l, t);
    return id + kp;
}
SP_HD constexpr int fwd_nchunks(int prec) { return fwd_chunk_id(prec, N_LAYERS, 0, 0, 0); }
SP_HD constexpr Chunk fwd_chunk(int prec, int id) {
    for (int l = 0; l < N_LAYERS; ++l)
        for (int g = 0; g < fwd_ngroups(prec, l); ++g)
            for (int s = 0; s < layer_nseg(l); ++s)
                for (int kp = 0; kp < fwd_seg_nparts(prec, l, s); ++kp)
                    if (fwd_chunk_id(prec, l, g, s, kp) == id) {
                        int G = group_g(prec), mb0 = g * G;
                        int nmb = layer_out_mb(l) - mb0 < G ? layer_out_mb(l) - mb0 : G;
                        int tot = seg_nks(prec, layer_seg_kind(l, s)), ks0 = kp * nks_max(prec);
                        int nks = tot - ks0 < nks_max(prec) ? tot - ks0 : nks_max(prec);
                        return Chunk{l, s, mb0, nmb, ks0, nks};
                    }
    return Chunk{-1, 0, 0, 0, 0, 0};
}

// ------------------------------------------------------------------ backward (dgrad) stream
// A = W^T: M space = C-rows of an input segment, K space = the layer's output vector.
// for l in 9..0: for s in segments: for g in groups(segment m-blocks): for part: chunk.
// Parts split the K range into <= nks_max pieces; layer 7 has one extra part for the
// raw-sigma slot that follows the 256 feature slots.
SP_HD constexpr int bwd_out_nks(int prec, int l) {
    return l < 8 ? 128 / kj_of(prec) : l == 8 ? 64 / kj_of(prec) : (kj_of(prec) == 8 ? 1 : 3);   // dz: q = 0,1,2
}
SP_HD constexpr int bwd_nparts(int prec, int l) { return cdiv(bwd_out_nks(prec, l), nks_max(prec)) + (l == 7 ? 1 : 0); }
SP_HD constexpr int bwd_part_ks0(int prec, int l, int part) { return part * nks_max(prec) < bwd_out_nks(prec, l) ? part * nks_max(prec) : bwd_out_nks(prec, l); }
SP_HD constexpr int bwd_part_nks(int prec, int l, int part) {
    int ks0 = part * nks_max(prec), tot = bwd_out_nks(prec, l);
    return ks0 >= tot ? 1 /* sigma slot */ : (tot - ks0 < nks_max(prec) ? tot - ks0 : nks_max(prec));
}
SP_HD constexpr int bwd_seg_ngroups(int prec, int l, int s) { return cdiv(vk_width(layer_seg_kind(l, s)) / 32, group_g(prec)); }
SP_HD constexpr int bwd_chunk_id(int prec, int l, int s, int g, int part) {
    int id = 0;
    for (int i = N_LAYERS - 1; i > l; --i)
        for (int t = 0; t < layer_nseg(i); ++t) id += bwd_seg_ngroups(prec, i, t) * bwd_nparts(prec, i);
    for (int t = 0; t < s; ++t) id += bwd_seg_ngroups(prec, l, t) * bwd_nparts(prec, l);
    return id + g * bwd_nparts(prec, l) + part;
}
SP_HD constexpr int bwd_nchunks(int prec) { return bwd_chunk_id(prec, -1, 0, 0, 0); }
SP_HD constexpr Chunk bwd_chunk(int prec, int id) {
    for (int l = N_LAYERS - 1; l >= 0; --l)
        for (int s = 0; s < layer_nseg(l); ++s)
            for (int g = 0; g < bwd_seg_ngroups(prec, l, s); ++g)
                for (int p = 0; p < bwd_nparts(prec, l); ++p)
                    if (bwd_chunk_id(prec, l, s, g, p) == id) {
                        int G = group_g(prec), mb0 = g * G, tot = vk_width(layer_seg_kind(l, s)) / 32;
                        int nmb = tot - mb0 < G ? tot - mb0 : G;
                        return Chunk{l, s, mb0, nmb, bwd_part_ks0(prec, l, p), bwd_part_nks(prec, l, p)};
                    }
    return Chunk{-1, 0, 0, 0, 0, 0};
}

// chunks that only the pose-gradient variant of the dgrad kernel consumes (gradients
// w.r.t. the encoded point x0 and the encoded view direction)
SP_HD constexpr bool bwd_chunk_optional(int prec, int id) {
    Chunk c = bwd_chunk(prec, id);
    return (c.layer == 8 && c.seg == 1) || (c.layer == 4 && c.seg == 1) || c.layer == 0;
}
SP_HD constexpr int bwd_next_id(int prec, int id, bool pose) {
    int n = (id + 1) % bwd_nchunks(prec);
    while (!pose && bwd_chunk_optional(prec, n)) n = (n + 1) % bwd_nchunks(prec);
    return n;
}

// byte offset of a chunk inside its stream (compile-time in device code)
SP_HD constexpr int64_t fwd_chunk_off(int prec, int id) {
    int64_t o = 0;
    for (int i = 0; i < id; ++i) o += chunk_bytes(prec, fwd_chunk(prec, i));
    return o;
}
SP_HD constexpr int64_t bwd_chunk_off(int prec, int id) {
    int64_t o = 0;
    for (int i = 0; i < id; ++i) o += chunk_bytes(prec, bwd_chunk(prec, i));
    return o;
}
SP_HD constexpr int64_t fwd_stream_bytes(int prec) { return fwd_chunk_off(prec, fwd_nchunks(prec)); }
SP_HD constexpr int64_t bwd_stream_bytes(int prec) { return bwd_chunk_off(prec, bwd_nchunks(prec)); }

// packed forward bias: [layer][m-block][half][16] floats (accumulator initial values)
SP_HD constexpr int bias_pk_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += layer_out_mb(i) * 32;
    return o;
}
enum { BIAS_PK_FLOATS = (7 * 8 + 9 + 4 + 1) * 32 };
// Raw-coordinate columns of the two layers that read the encoded point (layer 0, skip layer 4), in the
// packed-bias order: [which: 0 = layer 0, 1 = layer 4][coord x, y, z][m-block 0..7][half][16] fp32.
// Build option -DSP_XYZ_EXACT=1 (bf16x3 only): with inverse-depth sampling (renderer.py:413-416) the sample point reaches
// |p| ~ 1e8 and a 16-bit-mantissa (head + tail) product of it is wrong by ~1e3 absolute; the forward kernels then take
// these three columns out of the MFMA stream (zero weights there) and start the accumulators at
// b + w_x p_x + w_y p_y + w_z p_z in fp32 FMAs.  Measured (round 3, same box): per-sample outputs at |p| ~ 1e8 1.2e-2 ->
// 2.7e-3, but the RENDERED outputs stay at 3e-5 ... 1.3e-4 (the huge activations of every later layer lose the same
// 8 bits), for +2.6 % forward time -- so the default build leaves it off and the mode runs inverse-depth passes on the
// fp32 kernels instead (frequency_nerf.get_precision).  The table is part of the packed blob either way.
enum { XYZ_PK_FLOATS = 2 * 3 * 256, AUX_PK_FLOATS = BIAS_PK_FLOATS + XYZ_PK_FLOATS };
#ifndef SP_XYZ_EXACT
#define SP_XYZ_EXACT 0
#endif
SP_HD constexpr bool xyz_exact(int prec) { return SP_XYZ_EXACT && prec == PREC_X3; }
SP_HD constexpr int xyz_pk_off(int which, int coord) { return BIAS_PK_FLOATS + (which * 3 + coord) * 256; }

// device blob produced by sparf_pack_weights for one network:
//   [fwd stream][bwd stream][bias_pk floats][xyz_pk floats]
SP_HD constexpr int64_t packed_fwd_off(int prec) { return 0; }
SP_HD constexpr int64_t packed_bwd_off(int prec) { return fwd_stream_bytes(prec); }
SP_HD constexpr int64_t packed_bias_off(int prec) { return fwd_stream_bytes(prec) + bwd_stream_bytes(prec); }
SP_HD constexpr int64_t packed_bytes(int prec) { return packed_bias_off(prec) + AUX_PK_FLOATS * 4; }
// The band weights of the BARF coarse-to-fine mask are NOT part of the blob: they depend on the
// `progress` scalar, which trainers rewrite through `.data` without touching a weight
// (nerf_trainer.py:273-275), so every pass gets its own 16-float vector (10 point bands, 4 view
// bands, 2 pad) from sparf_c2f_weights and the backward reads the vector its forward used.
enum { C2F_FLOATS = 16 };

// host-built int32 gather tables (static per precision), uploaded once by the caller:
//   [fwd stream elements][bwd stream elements][bias_pk][xyz_pk][wgrad source per parameter]
SP_HD constexpr int64_t tbl_fwd_off(int prec) { return 0; }
SP_HD constexpr int64_t tbl_bwd_off(int prec) { return fwd_stream_bytes(prec) / abytes_of(prec); }
SP_HD constexpr int64_t tbl_bias_off(int prec) { return tbl_bwd_off(prec) + bwd_stream_bytes(prec) / abytes_of(prec); }
SP_HD constexpr int64_t tbl_wsrc_off(int prec) { return tbl_bias_off(prec) + AUX_PK_FLOATS; }
SP_HD constexpr int64_t tbl_count(int prec) { return tbl_wsrc_off(prec) + N_PARAMS; }

}  // namespace sparf
