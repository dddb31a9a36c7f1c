// Accuracy of device sincosf over the argument range the positional encoding sees with inverse-depth
// sampling (|p * 2^k * pi| up to ~1e11), against sin/cos in double of the same fp32 argument.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on tools/probes/sincos_probe.hip -o /tmp/sincos_probe && /tmp/sincos_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincosf(x[i], &s[i], &c[i]);
}

int main() {
    const int n = 1 << 16;
    for (double scale : {1.0, 1e2, 1e4, 1e6, 1e8, 1e10, 1e12}) {
        std::vector<float> x(n), s(n), c(n);
        unsigned long long st = 88172645463325252ull;
        for (int i = 0; i < n; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            x[i] = (float)(((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0) * scale);
        }
        float *dx, *ds, *dc;
        hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
        hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double es = 0, ec = 0;
        for (int i = 0; i < n; ++i) {
            es = fmax(es, fabs((double)s[i] - sin((double)x[i])));
            ec = fmax(ec, fabs((double)c[i] - cos((double)x[i])));
        }
        printf("|x| <= %g: max abs err sin %.2e cos %.2e\n", scale, es, ec);
        hipFree(dx); hipFree(ds); hipFree(dc);
    }
    return 0;
}
