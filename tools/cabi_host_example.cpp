// A host that uses libsparf_hip.so through its C ABI only (include/sparf_hip.h): no Python, no torch.
// One network pass forward + backward on rays read from a file; what INTEGRATION.md section B describes,
// as a program.  tests/test_cabi_host_gpu.py builds it, feeds it seeded inputs and checks its outputs
// and gradients against the oracle.
//
//   hipcc -std=c++17 -O2 tools/cabi_host_example.cpp -Iinclude -Lsparf_amd -lsparf_hip -Wl,-rpath,$PWD/sparf_amd -o tools/cabi_host_example.out
//   tools/cabi_host_example.out in.bin out.bin
//
// in.bin : int32 {prec, nrays, nsamp, white_bg}, then float32: the 20 parameter tensors W0,b0,...,W9,b9 in nn.Linear
//          layout (SPARF_N_PARAMS values), center[nrays][3], dir[nrays][3], t[nrays][nsamp],
//          g_rgb[nrays][3], g_depth[nrays]
// out.bin: float32: rgb[nrays][3], depth[nrays], opacity[nrays], weights[nrays][nsamp], grad_params[SPARF_N_PARAMS],
//          d_center[nrays][3], d_dir[nrays][3]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sparf_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_SP(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s returned %d\n", #x, r_); return 3; } } while (0)

static const int kOut[SPARF_N_LAYERS] = {256, 256, 256, 256, 256, 256, 256, 257, 128, 3};
static const int kIn[SPARF_N_LAYERS] = {63, 256, 256, 256, 319, 256, 256, 256, 283, 128};

template <class T> static T* dev_alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); std::exit(2); }
    return (T*)p;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    if (sparf_abi_version() != SPARF_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4) return 1;
    const int prec = hdr[0], R = hdr[1], N = hdr[2], white_bg = hdr[3];
    const size_t rows = (size_t)R * N;
    std::vector<float> params(SPARF_N_PARAMS), center(R * 3), dir(R * 3), t(rows), g_rgb(R * 3), g_depth(R);
    auto rd = [&](std::vector<float>& v) { return std::fread(v.data(), 4, v.size(), f) == v.size(); };
    if (!rd(params) || !rd(center) || !rd(dir) || !rd(t) || !rd(g_rgb) || !rd(g_depth)) return 1;
    std::fclose(f);

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    // parameters: one device buffer, 20 pointers into it
    float* d_params = dev_alloc<float>(SPARF_N_PARAMS);
    CHECK_HIP(hipMemcpy(d_params, params.data(), params.size() * 4, hipMemcpyHostToDevice));
    const float* ptrs[2 * SPARF_N_LAYERS];
    size_t off = 0;
    for (int l = 0; l < SPARF_N_LAYERS; ++l) {
        ptrs[2 * l] = d_params + off; off += (size_t)kOut[l] * kIn[l];
        ptrs[2 * l + 1] = d_params + off; off += kOut[l];
    }
    if (off != SPARF_N_PARAMS) return 1;
    // static tables (host) -> device, packed weights, band weights (no c2f: all ones)
    std::vector<int32_t> tables(sparf_table_count(prec));
    CHECK_SP(sparf_build_tables(prec, tables.data()));
    int32_t* d_tables = dev_alloc<int32_t>(tables.size());
    CHECK_HIP(hipMemcpy(d_tables, tables.data(), tables.size() * 4, hipMemcpyHostToDevice));
    char* d_packed = dev_alloc<char>(sparf_packed_bytes(prec));
    CHECK_SP(sparf_pack_weights(prec, ptrs, d_tables, d_packed, stream));
    float* d_c2f = dev_alloc<float>(16);
    CHECK_SP(sparf_c2f_weights(nullptr, 0, 0.0f, 1.0f, d_c2f, stream));

    auto up = [&](const std::vector<float>& v) { float* p = dev_alloc<float>(v.size()); hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
    float *d_center = up(center), *d_dir = up(dir), *d_t = up(t), *d_grgb = up(g_rgb), *d_gdepth = up(g_depth);

    sparf_pass_fwd_t fa = {};
    fa.prec = prec; fa.nrays = R; fa.nsamp = N;
    fa.center = d_center; fa.dir = d_dir; fa.t = d_t; fa.noise = nullptr; fa.noise_scale = 0.0f; fa.white_bg = white_bg;
    fa.packed = d_packed; fa.c2f = d_c2f;
    fa.save = dev_alloc<char>(sparf_save_bytes(prec, (int64_t)rows));
    fa.venc_ws = dev_alloc<char>((size_t)R * 32 * (prec == SPARF_PREC_BF16 ? 2 : 4));
    fa.raylen = dev_alloc<float>(R); fa.sigma_raw = dev_alloc<float>(rows); fa.rgb_samples = dev_alloc<float>(rows * 3);
    fa.density = dev_alloc<float>(rows); fa.weights = dev_alloc<float>(rows); fa.rgb = dev_alloc<float>(R * 3);
    fa.depth = dev_alloc<float>(R); fa.opacity = dev_alloc<float>(R); fa.depth_var = dev_alloc<float>(R);
    fa.rgb_var = dev_alloc<float>(R); fa.all_cumulated = dev_alloc<float>(R);
    CHECK_SP(sparf_pass_forward(&fa, stream));

    sparf_pass_bwd_t ba = {};
    ba.prec = prec; ba.nrays = R; ba.nsamp = N;
    ba.center = d_center; ba.dir = d_dir; ba.t = d_t; ba.noise = nullptr; ba.noise_scale = 0.0f; ba.white_bg = white_bg;
    ba.packed = d_packed; ba.c2f = d_c2f; ba.tables = d_tables; ba.save = fa.save;
    ba.raylen = fa.raylen; ba.sigma_raw = fa.sigma_raw; ba.rgb_samples = fa.rgb_samples; ba.weights = fa.weights;
    ba.g_rgb = d_grgb; ba.g_depth = d_gdepth; ba.g_opacity = nullptr; ba.g_weights = nullptr;
    ba.ws = dev_alloc<char>(sparf_bwd_workspace_bytes(prec, R, N, 1));
    ba.grad_params = dev_alloc<float>(SPARF_N_PARAMS);
    ba.d_center = dev_alloc<float>(R * 3); ba.d_dir = dev_alloc<float>(R * 3);
    CHECK_SP(sparf_pass_backward(&ba, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    auto down = [&](const float* p, size_t n) { std::vector<float> h(n); hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost); std::fwrite(h.data(), 4, n, o); };
    down(fa.rgb, R * 3); down(fa.depth, R); down(fa.opacity, R); down(fa.weights, rows);
    down(ba.grad_params, SPARF_N_PARAMS); down(ba.d_center, R * 3); down(ba.d_dir, R * 3);
    std::fclose(o);
    std::printf("cabi_host_example: prec %d, %d rays x %d samples, forward + backward ok\n", prec, R, N);

    // ---- ray segments (sparf_segment_t): the same pass as TWO render calls laid back to back.
    // (a) both segments carry their own upstream gradients -> bit-identical to the unsegmented backward;
    // (b) only the second one does -> equal to an unsegmented backward whose first rays have zero gradients, while
    //     the kernels only run over the second segment's rows (R0 * N is a multiple of 32).
    {
        auto down_v = [&](const float* p, size_t n) { std::vector<float> h(n); hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost); return h; };
        auto maxdiff = [](const std::vector<float>& a, const std::vector<float>& b) { float m = 0.f, s = 0.f; for (size_t i = 0; i < a.size(); ++i) { m = std::fmax(m, std::fabs(a[i] - b[i])); s = std::fmax(s, std::fabs(b[i])); } return m / (s + 1e-30f); };
        const int R0 = (R / 2) / 4 * 4;
        const std::vector<float> ref_gp = down_v(ba.grad_params, SPARF_N_PARAMS), ref_dc = down_v(ba.d_center, (size_t)R * 3);
        sparf_segment_t seg[2] = {{0, R0, 0.0f, d_grgb, d_gdepth, nullptr, nullptr}, {R0, R - R0, 0.0f, d_grgb + (size_t)R0 * 3, d_gdepth + R0, nullptr, nullptr}};
        sparf_pass_bwd_t bs = ba;
        bs.g_rgb = bs.g_depth = nullptr; bs.nseg = 2; bs.seg = seg;
        bs.grad_params = dev_alloc<float>(SPARF_N_PARAMS); bs.d_center = dev_alloc<float>(R * 3); bs.d_dir = dev_alloc<float>(R * 3);
        CHECK_SP(sparf_pass_backward(&bs, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        const float e_a = std::fmax(maxdiff(down_v(bs.grad_params, SPARF_N_PARAMS), ref_gp), maxdiff(down_v(bs.d_center, (size_t)R * 3), ref_dc));
        // (b): reference = unsegmented backward with the first R0 rays' gradients zeroed
        std::vector<float> z_rgb(g_rgb), z_depth(g_depth);
        for (int i = 0; i < R0 * 3; ++i) z_rgb[i] = 0.f;
        for (int i = 0; i < R0; ++i) z_depth[i] = 0.f;
        sparf_pass_bwd_t bz = ba;
        bz.g_rgb = up(z_rgb); bz.g_depth = up(z_depth);
        bz.grad_params = dev_alloc<float>(SPARF_N_PARAMS); bz.d_center = dev_alloc<float>(R * 3); bz.d_dir = dev_alloc<float>(R * 3);
        CHECK_SP(sparf_pass_backward(&bz, stream));
        seg[0].g_rgb = seg[0].g_depth = nullptr;
        CHECK_SP(sparf_pass_backward(&bs, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        const float e_b = std::fmax(maxdiff(down_v(bs.grad_params, SPARF_N_PARAMS), down_v(bz.grad_params, SPARF_N_PARAMS)),
                                    maxdiff(down_v(bs.d_center, (size_t)R * 3), down_v(bz.d_center, (size_t)R * 3)));
        std::printf("cabi_host_example: segments: both active %.2e (bit-identical expected), second only %.2e\n", e_a, e_b);
        if (e_a != 0.0f || !(e_b < 5e-4f)) return 3;
    }
    return 0;
}
