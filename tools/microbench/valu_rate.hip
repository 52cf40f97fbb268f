// valu_rate.hip -- how many shader cycles does one wave64 FP32 vector instruction occupy a gfx950 SIMD?
// (round 5: the roofline arithmetic of the level-0 evaluation depends on it -- 2 cycles (SIMD-32) or 4 (SIMD-16 + packed))
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_rate valu_rate.hip && ./valu_rate
// Every thread runs `iters` trips of 16 independent operations; W waves per SIMD resident on every CU.  Reports, per kind,
// wave-instructions per SIMD and shader cycle (s_memtime) -- 0.5 means two cycles per instruction -- and the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters, unsigned long long *clk) {
  float acc[16];
  f2 pacc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3f + i, pacc[i] = f2{acc[i], acc[i] + 1};
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll
  for (int i = 0; i < 16; i++) asm volatile("" : "+v"(acc[i]));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) acc[i] = __builtin_fmaf(acc[i], a, b);              // v_fma_f32
      if (KIND == 1) acc[i] = acc[i] * a;                                // v_mul_f32
      if (KIND == 2) acc[i] = acc[i] + a;                                // v_add_f32
      if (KIND == 3) pacc[i] = __builtin_elementwise_fma(pacc[i], f2{a, a}, f2{b, b}); // v_pk_fma_f32
      if (KIND == 4) pacc[i] = pacc[i] * f2{a, b};                       // v_pk_mul_f32
      if (KIND == 5) acc[i] = __builtin_amdgcn_rcpf(acc[i]);             // v_rcp_f32 (transcendental rate)
      if (KIND == 6) acc[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[i]), 0xB1, 0xf, 0xf, false)) + acc[i]; // v_add with DPP
      if (KIND == 7) acc[i] = acc[i] > a ? acc[i] - b : acc[i];          // v_cmp + v_cndmask + v_sub
      if (KIND == 8) acc[i] = __builtin_fmaf(acc[(i + 1) & 15], acc[(i + 5) & 15], acc[i]); // v_fma_f32, three distinct VGPR sources
      if (KIND == 9) pacc[i] = __builtin_elementwise_fma(pacc[(i + 1) & 15], pacc[(i + 5) & 15], pacc[i]); // v_pk_fma_f32, three distinct VGPR pairs
      if (KIND == 10) pacc[i] = pacc[i] + pacc[(i + 3) & 15];            // v_pk_add_f32
      if (KIND == 11) pacc[i] = __builtin_elementwise_fma(pacc[i], pacc[i], pacc[i]); // v_pk_fma_f32, one VGPR pair for all three sources
    }
  }
  #pragma unroll
  for (int i = 0; i < 16; i++) asm volatile("" : "+v"(acc[i]), "+v"(pacc[i]));
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i] + pacc[i].x + pacc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, iters = 32768;
  const char *names[12] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_rcp_f32", "v_add_f32 dpp", "cmp+cndmask+sub (3 instr)", "v_fma_f32 3 vgpr sources", "v_pk_fma_f32 3 vgpr pairs", "v_pk_add_f32", "v_pk_fma_f32 same pair x3"};
  printf("{\"cus\": %d, \"results\": [\n", cus);
  for (int waves = 2; waves <= 8; waves *= 2) {
    const int grid = cus * waves; // 256 threads = one wave per SIMD per workgroup
    float *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(float) * grid * 256));
    CHECK(hipMalloc(&clk, 16 * grid));
    for (int kind = 0; kind < 12; kind++) {
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      auto launch = [&]() {
        switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        default: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk); break;
        }
      };
      launch();
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 5; r++) launch();
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      std::vector<unsigned long long> h(2 * grid);
      CHECK(hipMemcpy(h.data(), clk, 16 * grid, hipMemcpyDeviceToHost));
      double cyc = 0, real = 0;
      for (int i = 0; i < grid; i++) cyc += h[2 * i], real += h[2 * i + 1];
      cyc /= grid, real /= grid;
      const double instr_per_wave = 16.0 * iters * (kind == 7 ? 3 : 1);
      // per SIMD: `waves` resident waves share it; cycles of the loop measured by wave 0 of each workgroup
      const double ghz = cyc / (real * 10.0) /* memrealtime: 100 MHz */, lane_ops = instr_per_wave * 64.0 * 4 * waves * cus / (ms * 1e-3);
      printf(" {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"clock_GHz\": %.3f, \"lane_ops_per_s_T\": %.2f, \"wall_cycles_per_wave_instr_per_simd\": %.2f}%s\n",
             names[kind], waves, ms, ghz, lane_ops / 1e12, 64.0 / (lane_ops / (4.0 * cus * ghz * 1e9)), (waves == 8 && kind == 11) ? "" : ",");
    }
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
  }
  printf("]}\n");
  return 0;
}
