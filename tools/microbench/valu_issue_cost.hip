// valu_issue_cost.hip -- shader cycles one wave64 vector instruction occupies its gfx950 SIMD, per opcode / encoding / operand kind.
// Round 5 finding (valu_rate.hip): `v_fmac_f32 v, v, v` issues in ~2.4 cycles, `v_fma_f32 v, s, v, v` / `v_add_f32 v, s, v` in ~4.4 --
// the cost of an instruction depends on its form.  This prices the forms the evaluation loop uses.
//   hipcc --offload-arch=gfx950 -O3 -o _valu_issue_cost valu_issue_cost.hip && ./_valu_issue_cost
// Every kernel: 8 waves per SIMD on every CU, `iters` trips of 16 independent copies of ONE instruction (inline asm, no dependences
// between the copies); cycles = wall time x clock (s_memtime / s_memrealtime inside the kernel) / instructions per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(NAME, ASM)                                                                                             \
  __global__ __launch_bounds__(256) void NAME(float *out, float sa, float sb, int iters, unsigned long long *clk) {   \
    float d[16], x[16], y[16];                                                                                        \
    for (int i = 0; i < 16; i++) d[i] = threadIdx.x * 1e-3f + i, x[i] = d[i] * 0.5f + 1.0f, y[i] = d[i] * 0.25f + 2.0f; \
    unsigned long long t0, r0, t1, r1, m = 0x5555555555555555ull;                                                      \
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");            \
    for (int it = 0; it < iters; it++) {                                                                              \
      _Pragma("unroll") for (int i = 0; i < 16; i++)                                                                  \
          asm volatile(ASM : "+v"(d[i]) : "v"(x[i]), "v"(y[(i + 3) & 15]), "s"(sa), "s"(sb), "s"(m) : "vcc");        \
    }                                                                                                                 \
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");            \
    float s = 0;                                                                                                      \
    for (int i = 0; i < 16; i++) s += d[i];                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                                          \
    if (threadIdx.x == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;                           \
  }

// %0 dest (VGPR, read-write), %1 %2 VGPR sources, %3 %4 SGPR floats, %5 SGPR pair (lane mask)
KERNEL(k_mul_vv, "v_mul_f32_e32 %0, %1, %2")
KERNEL(k_mul_sv, "v_mul_f32_e32 %0, %3, %2")
KERNEL(k_mul_lit, "v_mul_f32_e32 %0, 0x40490fdb, %2")
KERNEL(k_mul_inl, "v_mul_f32_e32 %0, 2.0, %2")
KERNEL(k_mul_e64_neg, "v_mul_f32_e64 %0, %1, -%2")
KERNEL(k_add_vv, "v_add_f32_e32 %0, %1, %2")
KERNEL(k_add_sv, "v_add_f32_e32 %0, %3, %2")
KERNEL(k_sub_vv, "v_sub_f32_e32 %0, %1, %2")
KERNEL(k_sub_inl, "v_sub_f32_e32 %0, 1.0, %2")
KERNEL(k_fmac_vv, "v_fmac_f32_e32 %0, %1, %2")
KERNEL(k_fmac_sv, "v_fmac_f32_e32 %0, %3, %2")
KERNEL(k_fma_vvv, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_fma_vvv_d, "v_fma_f32 %0, %1, %2, %1")
KERNEL(k_fma_svv, "v_fma_f32 %0, %3, %2, %1")
KERNEL(k_fma_neg, "v_fma_f32 %0, -%1, %2, %1")
KERNEL(k_fma_inl, "v_fma_f32 %0, %1, %2, 1.0")
KERNEL(k_cndmask_vcc, "v_cndmask_b32_e32 %0, %1, %2, vcc")
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %1, %2, %5")
KERNEL(k_cndmask_0, "v_cndmask_b32_e64 %0, 0, %2, %5")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32_e32 vcc, %1, %2")
KERNEL(k_cmp_s, "v_cmp_lt_f32_e64 s[20:21], %1, %2")
KERNEL(k_cmp_sv, "v_cmp_lt_f32_e32 vcc, %3, %2")
KERNEL(k_cmp_abs, "v_cmp_gt_f32_e64 s[20:21], |%1|, %3")
KERNEL(k_cmp_class, "v_cmp_class_f32_e64 s[20:21], %1, %3")
KERNEL(k_mov, "v_mov_b32_e32 %0, %1")
KERNEL(k_cvt_i32, "v_cvt_i32_f32_e32 %0, %1")
KERNEL(k_fract, "v_fract_f32_e32 %0, %1")
KERNEL(k_floor, "v_floor_f32_e32 %0, %1")
KERNEL(k_rcp, "v_rcp_f32_e32 %0, %1")
KERNEL(k_frexp, "v_frexp_exp_i32_f32_e32 %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %1, %2")
KERNEL(k_mul_lo_s, "v_mul_lo_u32 %0, %3, %2")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %1")
KERNEL(k_mad_u24_s, "v_mad_u32_u24 %0, %3, %2, %1")
KERNEL(k_and_vv, "v_and_b32_e32 %0, %1, %2")
KERNEL(k_and_lit, "v_and_b32_e32 %0, 0x7fffffff, %2")
KERNEL(k_lshl, "v_lshlrev_b32_e32 %0, 2, %2")
KERNEL(k_add_u32, "v_add_u32_e32 %0, %1, %2")
KERNEL(k_add_u32_s, "v_add_u32_e32 %0, %3, %2")
KERNEL(k_min_i32_s, "v_min_i32_e32 %0, %3, %2")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %1, %2, 2")
KERNEL(k_med3, "v_med3_f32 %0, %1, %2, %0")
KERNEL(k_min_vv, "v_min_f32_e32 %0, %1, %2")
KERNEL(k_min_inl, "v_min_f32_e32 %0, 1.0, %2")
KERNEL(k_max_vv, "v_max_f32_e32 %0, %1, %2")
KERNEL(k_cndmask_m1, "v_cndmask_b32_e64 %0, 0, -1, %5")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32_e32 %0, %1")
KERNEL(k_mul_u24, "v_mul_u32_u24_e32 %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 2, %2")
KERNEL(k_or_vv, "v_or_b32_e32 %0, %1, %2")
KERNEL(k_xor_vv, "v_xor_b32_e32 %0, %1, %2")
KERNEL(k_sub_u32, "v_sub_u32_e32 %0, %1, %2")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
KERNEL(k_and_or, "v_and_or_b32 %0, %1, %2, %0")
KERNEL(k_max_abs, "v_max_f32_e64 %0, |%1|, |%2|")
KERNEL(k_add_e64_abs, "v_add_f32_e64 %0, |%1|, |%2|")
KERNEL(k_fmac_dep, "v_fmac_f32_e32 %0, %0, %2")
KERNEL(k_mul_dep_chain, "v_mul_f32_e32 %1, %1, %2")
KERNEL(k_dpp_add, "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane_pair, "v_readlane_b32 s20, %1, 3\n v_mov_b32_e32 %0, %2")

// a DEPENDENT chain: every instruction reads the previous one's result (issue-to-issue latency of one wave)
__global__ __launch_bounds__(256) void k_dep_mul(float *out, float sa, float sb, int iters, unsigned long long *clk) {
  float d = threadIdx.x * 1e-3f + 1.0f, y = 1.0000001f;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(d) : "v"(y));
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  out[blockIdx.x * 256 + threadIdx.x] = d;
  if (threadIdx.x == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}
__global__ __launch_bounds__(256) void k_dep_cmp_cnd(float *out, float sa, float sb, int iters, unsigned long long *clk) {
  float d = threadIdx.x * 1e-3f + 1.0f, y = 1.0000001f;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(d) : "v"(y) : "s20", "s21");
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  out[blockIdx.x * 256 + threadIdx.x] = d;
  if (threadIdx.x == 0) clk[2 * blockIdx.x] = t1 - t0, clk[2 * blockIdx.x + 1] = r1 - r0;
}
typedef void (*kfn)(float *, float, float, int, unsigned long long *);
struct Entry { const char *name; kfn fn; int n; };

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, iters = 16384;
  const int waves = argc > 1 ? atoi(argv[1]) : 8, grid = cus * waves;
  float *out;
  unsigned long long *clk;
  CHECK(hipMalloc(&out, sizeof(float) * grid * 256));
  CHECK(hipMalloc(&clk, 16 * grid));
#define E(k) {#k, k, 1}
  Entry es[] = {E(k_mul_vv), E(k_mul_sv), E(k_mul_lit), E(k_mul_inl), E(k_mul_e64_neg), E(k_add_vv), E(k_add_sv), E(k_sub_vv), E(k_sub_inl),
                E(k_fmac_vv), E(k_fmac_sv), E(k_fma_vvv), E(k_fma_vvv_d), E(k_fma_svv), E(k_fma_neg), E(k_fma_inl), E(k_cndmask_vcc), E(k_cndmask_s),
                E(k_cndmask_0), E(k_cmp_vcc), E(k_cmp_s), E(k_cmp_sv), E(k_cmp_abs), E(k_cmp_class), E(k_mov), E(k_cvt_i32), E(k_fract), E(k_floor),
                E(k_rcp), E(k_frexp), E(k_mul_lo), E(k_mul_lo_s), E(k_mad_u24), E(k_mad_u24_s), E(k_and_vv), E(k_and_lit), E(k_lshl), E(k_add_u32),
                E(k_add_u32_s), E(k_min_i32_s), E(k_add_lshl), E(k_med3), E(k_min_vv), E(k_min_inl), E(k_max_vv), E(k_cndmask_m1), E(k_cvt_f32_i32), E(k_mul_u24),
                E(k_lshl_add), E(k_or_vv), E(k_xor_vv), E(k_sub_u32), E(k_bfi), E(k_and_or), E(k_max_abs), E(k_add_e64_abs), E(k_fmac_dep), E(k_dpp_add),
                {"k_readlane_pair", k_readlane_pair, 2}, {"k_dep_mul (dependent chain)", k_dep_mul, 1}, {"k_dep_cmp_cnd (dependent v_cmp -> v_cndmask pairs)", k_dep_cmp_cnd, 1}};
  printf("{\"cus\": %d, \"waves_per_simd\": %d, \"results\": {\n", cus, waves);
  const int ne = sizeof(es) / sizeof(es[0]);
  for (int e = 0; e < ne; e++) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(es[e].fn, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(es[e].fn, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::vector<unsigned long long> h(2 * grid);
    CHECK(hipMemcpy(h.data(), clk, 16 * grid, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; i++) cyc += h[2 * i], real += h[2 * i + 1];
    const double ghz = cyc / (real * 10.0);
    const double instr_per_simd = 16.0 * iters * waves * es[e].n; // waves resident per SIMD, each issuing 16 x iters instructions
    printf("  \"%s\": {\"ms\": %.3f, \"clock_GHz\": %.3f, \"cycles_per_instruction\": %.2f}%s\n", es[e].name + 2, ms, ghz, ms * 1e-3 * ghz * 1e9 / instr_per_simd,
           e + 1 < ne ? "," : "");
  }
  printf("}}\n");
  return 0;
}
