// Hardware probe: semantics of ds_read_b64_tr_b16 (gfx950).  Every lane supplies an 8-byte LDS
// address; prints, for each lane, which input elements came back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x, i = l & 15, g = l >> 4;
  // mode 0: contiguous 128 B per 16-lane group; mode 1: 4 rows of stride 288 elements, 16 features per group
  int addr = mode == 0 ? (i * 4 + g * 64) : ((g >> 1) * 8 + (i >> 2)) * 288 + (g & 1) * 16 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + addr));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 64>>>(d, mode);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int i = l & 15, g = l >> 4;
      int expect = mode == 0 ? (i + j * 16 + g * 64) : ((g >> 1) * 8 + j) * 288 + (g & 1) * 16 + i;
      if (h[l * 4 + j] != expect) { if (bad < 8) printf("mode %d lane %d j %d got %d expect %d\n", mode, l, j, h[l*4+j], expect); ++bad; }
    }
    printf("mode %d: %s (%d mismatches)\n", mode, bad ? "MISMATCH" : "OK", bad);
  }
  return 0;
}
