// A complete training loop of the hierarchical renderer through the C ABI only (include/sparf_hip.h) -- ray
// generation, stratified depths, coarse pass, inverse-CDF resampling + sort, fine pass, photometric loss,
// both backward passes, gradient clipping + Adam -- from a C++ host without Python or torch: the path of
// /root/reference/source/models/renderer.py:250-345 + nerf_trainer.py:181-185 as a sequence of extern "C" calls.
// tests/test_cabi_host_gpu.py feeds it the initial weights, cameras, target colours and the random draws of every
// step through a file, and compares the loss curve with the Python `Graph` path given the same draws.
//
//   hipcc -std=c++17 -O2 tools/cabi_train_example.cpp -Iinclude -Lsparf_amd -lsparf_hip -Wl,-rpath,$PWD/sparf_amd -o tools/cabi_train_example.out
//   tools/cabi_train_example.out in.bin out.bin
//
// in.bin : int32 {prec, nimg, width, rays_per_image R, Nc, Nf, steps}; float32 {dmin, dmax, lr, clip};
//          float32 params_coarse[SPARF_N_PARAMS], params_fine[SPARF_N_PARAMS], pose[nimg][3][4], intr[nimg][3][3];
//          per step: int64 ray_idx[R]; float32 jitter[nimg*R*Nc], grid[Nf+1], target[nimg*R*3]
// out.bin: float32 loss[steps]
#include <hip/hip_runtime.h>

#include <cstdint>
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
    hipMemset(p, 0, n * sizeof(T) + 256);
    return (T*)p;
}

struct Net {
    float* params;                       // SPARF_N_PARAMS, nn.Linear layout
    const float* ptrs[2 * SPARF_N_LAYERS];
    float *grad, *m, *v, *adam_ws, *norm;
    char* packed;
    void init(const std::vector<float>& h, int prec) {
        params = dev_alloc<float>(SPARF_N_PARAMS);
        hipMemcpy(params, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        size_t off = 0;
        for (int l = 0; l < SPARF_N_LAYERS; ++l) {
            ptrs[2 * l] = params + off; off += (size_t)kOut[l] * kIn[l];
            ptrs[2 * l + 1] = params + off; off += kOut[l];
        }
        grad = dev_alloc<float>(SPARF_N_PARAMS); m = dev_alloc<float>(SPARF_N_PARAMS); v = dev_alloc<float>(SPARF_N_PARAMS);
        adam_ws = dev_alloc<float>(sparf_adam_workspace_floats()); norm = dev_alloc<float>(1);
        packed = dev_alloc<char>(sparf_packed_bytes(prec));
    }
};

struct Pass {                            // buffers of one network pass over `rays` x `ns` samples
    sparf_pass_fwd_t f = {};
    sparf_pass_bwd_t b = {};
    void init(int prec, int rays, int ns, const Net& net, const float* center, const float* dir, const float* t, const float* c2f,
              const int32_t* tables, const float* g_rgb) {
        const size_t rows = (size_t)rays * ns;
        f.prec = prec; f.nrays = rays; f.nsamp = ns; f.center = center; f.dir = dir; f.t = t; f.noise = nullptr; f.noise_scale = 0.f; f.white_bg = 0;
        f.packed = net.packed; f.c2f = c2f; f.save = dev_alloc<char>(sparf_save_bytes(prec, (int64_t)rows));
        f.venc_ws = dev_alloc<char>((size_t)rays * 32 * (prec == SPARF_PREC_BF16 ? 2 : 4));
        f.raylen = dev_alloc<float>(rays); f.sigma_raw = dev_alloc<float>(rows); f.rgb_samples = dev_alloc<float>(rows * 3);
        f.density = dev_alloc<float>(rows); f.weights = dev_alloc<float>(rows); f.rgb = dev_alloc<float>(rays * 3);
        f.depth = dev_alloc<float>(rays); f.opacity = dev_alloc<float>(rays); f.depth_var = dev_alloc<float>(rays);
        f.rgb_var = dev_alloc<float>(rays); f.all_cumulated = dev_alloc<float>(rays);
        b.prec = prec; b.nrays = rays; b.nsamp = ns; b.center = center; b.dir = dir; b.t = t; b.noise = nullptr; b.noise_scale = 0.f; b.white_bg = 0;
        b.packed = net.packed; b.c2f = c2f; b.tables = tables; b.save = f.save;
        b.raylen = f.raylen; b.sigma_raw = f.sigma_raw; b.rgb_samples = f.rgb_samples; b.weights = f.weights;
        b.g_rgb = g_rgb; b.g_depth = nullptr; b.g_opacity = nullptr; b.g_weights = nullptr;
        b.ws = dev_alloc<char>(sparf_bwd_workspace_bytes(prec, rays, ns, 0));
        b.grad_params = net.grad; b.d_center = nullptr; b.d_dir = nullptr;
    }
};

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hi[7];
    float hf[4];
    if (std::fread(hi, 4, 7, f) != 7 || std::fread(hf, 4, 4, f) != 4) return 1;
    const int prec = hi[0], B = hi[1], W = hi[2], R = hi[3], Nc = hi[4], Nf = hi[5], steps = hi[6];
    const float dmin = hf[0], dmax = hf[1], lr = hf[2], clip = hf[3];
    const int rays = B * R, Nt = Nc + Nf;
    auto rdf = [&](size_t n) { std::vector<float> v(n); if (std::fread(v.data(), 4, n, f) != n) std::exit(1); return v; };
    std::vector<float> pc = rdf(SPARF_N_PARAMS), pf = rdf(SPARF_N_PARAMS), pose = rdf((size_t)B * 12), intr = rdf((size_t)B * 9);

    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    std::vector<int32_t> tables(sparf_table_count(prec));
    CHECK_SP(sparf_build_tables(prec, tables.data()));
    int32_t* d_tables = dev_alloc<int32_t>(tables.size());
    CHECK_HIP(hipMemcpy(d_tables, tables.data(), tables.size() * 4, hipMemcpyHostToDevice));
    Net coarse, fine;
    coarse.init(pc, prec);
    fine.init(pf, prec);
    float* d_c2f = dev_alloc<float>(16);
    CHECK_SP(sparf_c2f_weights(nullptr, 0, 0.f, 1.f, d_c2f, s));
    auto upf = [&](const std::vector<float>& v) { float* p = dev_alloc<float>(v.size()); hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice); return p; };
    float *d_pose = upf(pose), *d_intr = upf(intr);
    int64_t* d_idx = dev_alloc<int64_t>(R);
    float *d_center = dev_alloc<float>(rays * 3), *d_dir = dev_alloc<float>(rays * 3);
    float *d_jit = dev_alloc<float>((size_t)rays * Nc), *d_tc = dev_alloc<float>((size_t)rays * Nc), *d_tall = dev_alloc<float>((size_t)rays * Nt);
    float *d_umid = dev_alloc<float>(Nf), *d_target = dev_alloc<float>(rays * 3);
    float *d_loss = dev_alloc<float>(1), *d_grgb = dev_alloc<float>(rays * 3), *d_grgbf = dev_alloc<float>(rays * 3);
    float* d_lossws = dev_alloc<float>(sparf_photometric_workspace_floats());
    Pass pcs, pfn;
    pcs.init(prec, rays, Nc, coarse, d_center, d_dir, d_tc, d_c2f, d_tables, d_grgb);
    pfn.init(prec, rays, Nt, fine, d_center, d_dir, d_tall, d_c2f, d_tables, d_grgbf);

    std::vector<float> losses(steps);
    std::vector<int64_t> idx(R);
    for (int it = 0; it < steps; ++it) {
        if (std::fread(idx.data(), 8, R, f) != (size_t)R) return 1;
        std::vector<float> jit = rdf((size_t)rays * Nc), grid = rdf(Nf + 1), target = rdf((size_t)rays * 3), umid(Nf);
        for (int j = 0; j < Nf; ++j) umid[j] = 0.5f * (grid[j] + grid[j + 1]);                  // renderer.py:441
        CHECK_HIP(hipMemcpyAsync(d_idx, idx.data(), R * 8, hipMemcpyHostToDevice, s));
        CHECK_HIP(hipMemcpyAsync(d_jit, jit.data(), jit.size() * 4, hipMemcpyHostToDevice, s));
        CHECK_HIP(hipMemcpyAsync(d_umid, umid.data(), Nf * 4, hipMemcpyHostToDevice, s));
        CHECK_HIP(hipMemcpyAsync(d_target, target.data(), target.size() * 4, hipMemcpyHostToDevice, s));
        // weights changed (Adam): repack both networks
        CHECK_SP(sparf_pack_weights(prec, coarse.ptrs, d_tables, coarse.packed, s));
        CHECK_SP(sparf_pack_weights(prec, fine.ptrs, d_tables, fine.packed, s));
        // renderer.py:273-291 rays of the selected pixels; :383-419 stratified depths
        CHECK_SP(sparf_ray_gen_forward(d_pose, d_intr, nullptr, d_idx, 0, W, B, R, d_center, d_dir, s));
        CHECK_SP(sparf_sample_coarse(d_jit, 0.5f, nullptr, nullptr, dmin, dmax - dmin, 0, rays, Nc, d_tc, s));
        CHECK_SP(sparf_pass_forward(&pcs.f, s));                                                   // :304-309
        CHECK_SP(sparf_sample_fine(pcs.f.weights, d_tc, d_umid, nullptr, dmin, dmax, rays, Nc, Nf, nullptr, d_tall, s));   // :323-336
        CHECK_SP(sparf_pass_forward(&pfn.f, s));                                                   // :338-343
        // base_losses.py:303-311 MSE on rgb and rgb_fine, with its gradient seeds
        CHECK_SP(sparf_photometric_loss(pcs.f.rgb, pfn.f.rgb, d_target, (int64_t)rays * 3, 0, 0.5f, d_loss, d_grgb, d_grgbf, d_lossws, s));
        CHECK_SP(sparf_pass_backward(&pfn.b, s));
        CHECK_SP(sparf_pass_backward(&pcs.b, s));
        // base.py:96-97 clip per network, nerf_trainer.py:181-185 Adam
        CHECK_SP(sparf_adam_step(coarse.ptrs, coarse.grad, coarse.m, coarse.v, coarse.adam_ws, coarse.norm, lr, 0.9f, 0.999f, 1e-8f, it + 1, clip, s));
        CHECK_SP(sparf_adam_step(fine.ptrs, fine.grad, fine.m, fine.v, fine.adam_ws, fine.norm, lr, 0.9f, 0.999f, 1e-8f, it + 1, clip, s));
        CHECK_HIP(hipMemcpyAsync(&losses[it], d_loss, 4, hipMemcpyDeviceToHost, s));
        CHECK_HIP(hipStreamSynchronize(s));
    }
    std::fclose(f);
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    std::fwrite(losses.data(), 4, losses.size(), o);
    std::fclose(o);
    std::printf("cabi_train_example: %d steps, %d rays x (%d + %d), loss %.6f -> %.6f\n", steps, rays, Nc, Nf, losses.front(), losses.back());
    return 0;
}
