#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define G(p) ((const __attribute__((address_space(1))) void *)(p))
#define L(p) ((__attribute__((address_space(3))) void *)(p))
template <int BYTES> __global__ void k(const int *src, int *out, int stride_dw) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) lds[i] = -1;
  __syncthreads();
  if constexpr (BYTES == 4) __builtin_amdgcn_global_load_lds(G(src + lane * stride_dw), L(&lds[0]), 4, 0, 0);
  if constexpr (BYTES == 12) __builtin_amdgcn_global_load_lds(G(src + lane * stride_dw), L(&lds[0]), 12, 0, 0);
  if constexpr (BYTES == 16) __builtin_amdgcn_global_load_lds(G(src + lane * stride_dw), L(&lds[0]), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = lds[i];
}
int main() {
  const int N = 4096;
  std::vector<int> h(N);
  for (int i = 0; i < N; i++) h[i] = i;
  int *d, *o;
  hipMalloc(&d, N * 4);
  hipMalloc(&o, 320 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<int> r(320);
  for (int b : {4, 12, 16}) {
    for (int stride : {4, 7}) {
      if (b == 4) k<4><<<1, 64>>>(d, o, stride);
      if (b == 12) k<12><<<1, 64>>>(d, o, stride);
      if (b == 16) k<16><<<1, 64>>>(d, o, stride);
      hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
      printf("bytes=%d stride=%d:\n", b, stride);
      for (int i = 0; i < 320; i++) printf("%d%c", r[i], (i % 32 == 31) ? '\n' : ' ');
    }
  }
}
