#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *a, unsigned long long *out, int mode, int iters) {
  if (threadIdx.x != 0) return;
  unsigned *p = (mode & 1) ? a + 64 * blockIdx.x : a;  // bit0: private address per block
  unsigned long long t0 = wall_clock64();
  unsigned v = 0;
  for (int i = 0; i < iters; i++) {
    if (mode & 4) v += __hip_atomic_load(p + (v & 1), __ATOMIC_RELAXED, (mode & 2) ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
    else v += __hip_atomic_fetch_add(p + (v & 0), 1u, __ATOMIC_RELAXED, (mode & 2) ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long t1 = wall_clock64();
  out[blockIdx.x] = (t1 - t0) | ((unsigned long long)(v & 1) << 63);
}
int main() {
  unsigned *a; unsigned long long *out, h[2048];
  hipMalloc(&a, 4 * 64 * 2048); hipMemset(a, 0, 4 * 64 * 2048); hipMalloc(&out, 8 * 2048);
  const char *names[8] = {"fetch_add agent shared", "fetch_add agent private", "fetch_add wg shared", "fetch_add wg private", "load agent shared", "load agent private", "load wg shared", "load wg private"};
  for (int blocks : {1, 64, 1024})
    for (int mode = 0; mode < 8; mode++) {
      const int iters = 200;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, a, out, mode, iters);
      hipDeviceSynchronize();
      hipMemcpy(h, out, 8 * blocks, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < blocks; i++) s += (double)(h[i] & 0x7fffffffffffffffull);
      printf("blocks %4d %-26s %.0f ns per op\n", blocks, names[mode], s / blocks / iters * 10.0);
    }
  return 0;
}
