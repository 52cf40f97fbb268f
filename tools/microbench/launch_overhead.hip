// Host cost of enqueueing a kernel launch vs the device's cost of a dependent launch boundary:
// 2000 dependent launches of a trivial 256-workgroup kernel on one stream; host time until the last launch call returns,
// device time from first start to last end.   hipcc --offload-arch=gfx950 -O2 launch_overhead.hip -o /tmp/lo && /tmp/lo
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int *p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
  int *d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    const int N = 2000;
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, d);
    const auto t1 = std::chrono::steady_clock::now();
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("rep %d: host %.2f us per launch call, device %.2f us per dependent launch\n", rep,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N, 1e3 * ms / N);
  }
  return 0;
}
