// cu_mask_overlap.hip -- can a chain of short dependent kernels (the small pyramid levels' launches) run UNDER a chip-filling
// streaming kernel (a level-0 evaluation launch) without being stretched, if the two streams are given disjoint CU masks?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/cu_mask_overlap.hip -o /tmp/cumask && /tmp/cumask
// Prints the chain's duration alone, next to the heavy kernel without masks, and with masks (light: `L` CUs, heavy: the rest),
// and the heavy kernel's duration in each case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void heavy(const float4 *p, size_t per_block, float *out) {
  const float4 *q = p + (size_t)blockIdx.x * per_block;
  float acc = 0;
  for (size_t i = threadIdx.x; i < per_block; i += 256) {
    const float4 v = q[i];
    acc += v.x + v.y * v.z + v.w;
#pragma unroll
    for (int k = 0; k < 40; k++) acc = acc * 1.0001f + 0.5f; // some arithmetic per load, like the eval kernel
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void tiny(float *state, const float *src, int n) { // ~ one small-level evaluation + LM step
  const int b = blockIdx.x;
  float v = state[b];
  for (int i = threadIdx.x; i < n; i += 256) v += src[(size_t)b * n + i] * 1e-9f;
  __shared__ float s[256];
  s[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0;
    for (int i = 0; i < 256; i++) t += s[i];
    state[b] = t / 256.f;
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("CUs: %d\n", ncu);
  const size_t per_block = 4096, nblocks = 60000; // 64 KiB per block, 3.9 GB in all
  float4 *big;
  float *out, *state, *src;
  CK(hipMalloc(&big, per_block * nblocks * sizeof(float4)));
  CK(hipMemset(big, 0, per_block * nblocks * sizeof(float4)));
  CK(hipMalloc(&out, 64));
  const int nprob = 512, n = 2048;
  CK(hipMalloc(&state, nprob * 4));
  CK(hipMalloc(&src, (size_t)nprob * n * 4));
  CK(hipMemset(state, 0, nprob * 4));
  CK(hipMemset(src, 0, (size_t)nprob * n * 4));
  for (int light_cus : {0, 32, 64, 96}) {
    hipStream_t sh, sl;
    if (light_cus == 0) {
      CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
      CK(hipStreamCreateWithFlags(&sl, hipStreamNonBlocking));
    } else {
      const int words = (ncu + 31) / 32;
      std::vector<uint32_t> ml(words, 0), mh(words, 0);
      for (int c = 0; c < ncu; c++) (c < light_cus ? ml : mh)[c / 32] |= 1u << (c % 32);
      CK(hipExtStreamCreateWithCUMask(&sh, words, mh.data()));
      CK(hipExtStreamCreateWithCUMask(&sl, words, ml.data()));
    }
    hipEvent_t h0, h1, l0, l1;
    CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1)); CK(hipEventCreate(&l0)); CK(hipEventCreate(&l1));
    auto run_heavy = [&](int reps) { for (int r = 0; r < reps; r++) hipLaunchKernelGGL(heavy, dim3(nblocks), dim3(256), 0, sh, big, per_block, out); };
    auto run_chain = [&](int len) { for (int k = 0; k < len; k++) hipLaunchKernelGGL(tiny, dim3(nprob), dim3(256), 0, sl, state, src, n); };
    // warm up
    run_heavy(1); run_chain(10);
    CK(hipDeviceSynchronize());
    float th_alone, tl_alone, th_both, tl_both;
    CK(hipEventRecord(h0, sh)); run_heavy(4); CK(hipEventRecord(h1, sh)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&th_alone, h0, h1));
    CK(hipEventRecord(l0, sl)); run_chain(200); CK(hipEventRecord(l1, sl)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&tl_alone, l0, l1));
    CK(hipEventRecord(h0, sh)); CK(hipEventRecord(l0, sl));
    run_heavy(4); run_chain(200);
    CK(hipEventRecord(h1, sh)); CK(hipEventRecord(l1, sl));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&th_both, h0, h1)); CK(hipEventElapsedTime(&tl_both, l0, l1));
    printf("light CUs %3d (0 = no masks): heavy x4 alone %.3f ms, chain of 200 alone %.3f ms | together: heavy %.3f ms, chain %.3f ms\n", light_cus,
           th_alone, tl_alone, th_both, tl_both);
    hipStreamDestroy(sh); hipStreamDestroy(sl);
  }
  return 0;
}
