// Is  v_dot2c_f32_bf16  x += head . {-1, 0}  bit-identical to  x - float(head)  (head = bf16(x), round to nearest)?
// The bf16x3 epilogue (mlp_fwd_impl.h, SP_LO_DOT2) wants the tail of the head + tail split in one instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/dot2_probe.hip -o /tmp/dot2_probe && /tmp/dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* __restrict__ x, int n, unsigned* __restrict__ bad, float* __restrict__ worst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const float r0 = x0 - __builtin_bit_cast(float, hb << 16), r1 = x1 - __builtin_bit_cast(float, hb & 0xffff0000u);
    const float d0 = __builtin_amdgcn_fdot2_f32_bf16(h, bf16x2{(__bf16)-1.0f, (__bf16)0.0f}, x0, false);
    const float d1 = __builtin_amdgcn_fdot2_f32_bf16(h, bf16x2{(__bf16)0.0f, (__bf16)-1.0f}, x1, false);
    if (__builtin_bit_cast(unsigned, r0) != __builtin_bit_cast(unsigned, d0) && !(r0 == 0.0f && d0 == 0.0f)) { atomicAdd(bad, 1u); worst[0] = x0; worst[1] = r0; worst[2] = d0; }
    if (__builtin_bit_cast(unsigned, r1) != __builtin_bit_cast(unsigned, d1) && !(r1 == 0.0f && d1 == 0.0f)) { atomicAdd(bad, 1u); worst[0] = x1; worst[1] = r1; worst[2] = d1; }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {          // random bit patterns of finite floats over the whole exponent range + activation-like magnitudes
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        float f;
        memcpy(&f, &u, 4);
        if (!(f == f) || f > 3e38f || f < -3e38f) f = (float)rand() / RAND_MAX;
        if (i & 1) f = ((float)rand() / RAND_MAX - 0.3f) * 8.0f;
        h[i] = f;
    }
    float *d, *w; unsigned* bad;
    hipMalloc(&d, n * 4); hipMalloc(&w, 16); hipMalloc(&bad, 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4); hipMemset(w, 0, 16);
    probe<<<n / 2 / 256, 256>>>(d, n, bad, w);
    unsigned nb; float ww[3];
    hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(ww, w, 12, hipMemcpyDeviceToHost);
    printf("dot2 probe: %u mismatches of %d (example x=%g sub=%g dot2=%g)\n", nb, n, ww[0], ww[1], ww[2]);
    return 0;
}
