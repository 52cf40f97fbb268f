#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
  const int l = threadIdx.x;
  // A[i][k] = (i+1) , k-th column scaled by 10^k ; B[k][j] = (j+1) if k==0 else 0  -> D[i][j] = (i+1)*(j+1)  (k=0 only)
  const int i = l % 16, kk = l / 16;
  float a = (kk == 0) ? (float)(i + 1) : 0.f;
  float b = (kk == 0) ? (float)(100 * (i + 1)) : 0.f; // here "i" plays j
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
  float *d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // expected D[i][j] = (i+1) * 100*(j+1); print for each lane/reg the decoded (i,j)
  for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; r++) { int v = (int)h[l * 4 + r]; printf("lane %d reg %d: val %d -> i=%d j=%d\n", l, r, v, (v / 100) ? 0 : 0, 0); }
  int ok = 1;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { int i = 4 * (l / 16) + r, j = l % 16; if ((int)h[l * 4 + r] != (i + 1) * 100 * (j + 1)) ok = 0; }
  printf("layout D[4*(l/16)+r][l%%16]: %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
