// tracker_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the direct photometric hot path.
//
//   eval_kernel<MODE, ...>    fused calcResPose+calcGSSSEPose (TrackerAndScaler.cpp:699-852,
//                              640-697) or calcResScale+calcGSSSEScale (:1007-1172, :966-1005)
//                              for a batch of independent problems: grid = (chunks, problems).
//   lm_kernel                  per problem: fixed-order reduction of the chunk partials and one
//                              step of the Levenberg-Marquardt state machine of
//                              trackNewestCoarse (:451-638) / optimizeScale (:854-964).
//   template / pyramid helpers interleave, scaleCoarseDepthL0 (:329-336), makeImages.
//
// Numerics contract: compiled with -ffp-contract=off; every per-point
// value (warp, bounds test, bilinear taps, residual, Huber weight, cut-off test) is computed
// with the same IEEE float32 operation sequence as the CPU oracle, so the integer outputs
// (numTermsInE, numSaturated, warped count) are bit-exact.  Sums are reduced in a fixed tree
// (thread-sequential -> 16-lane DPP rows -> 16 rows sequential -> chunks sequential in
// double): deterministic run to run, equal to the reference's SSE lane order only up to
// float rounding (quirk Q4 is a CPU artefact, not reproduced).  No float atomics.
#include "dsm_kernels.hpp"
#include "lm_math.hpp"
#include "xwg_sync.hpp"


namespace dsm {


// ------------------------------------------------------------------------------------------
// wave64 helpers: DPP butterflies inside 16-lane rows (v_add_f32 ... quad_perm / row_mirror)
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// after this every lane holds the sum over its 16-lane row: the balanced tree ((l0 + l1) + (l2 + l3)) + ... over the row's lanes.
// (An add with a DPP operand costs 4.65 cycles where a plain v_add_f32 costs 2.5, and a chunk's 52 row sums are 970 vector cycles per
// wave -- 1.8 points' worth of the loop.  Round 5 tried the exchange through the LDS crossbar instead (ds_swizzle + plain add: the same
// tree, the same bits): level 0 5.2 -> 4.7 TB/s, 61.1 k -> 54.7 k frames/s -- four dependent LDS round trips per sum cost more than they
// save; profiles/r05_ab_swizzle_row_sums.log.)
__device__ __forceinline__ float row16_sum(float v) {
  v = v + dpp_f<0xB1>(v);  // quad_perm [1,0,3,2]
  v = v + dpp_f<0x4E>(v);  // quad_perm [2,3,0,1]
  v = v + dpp_f<0x141>(v); // row_half_mirror
  v = v + dpp_f<0x140>(v); // row_mirror
  return v;
}

// clang vector types: loads through address_space(1) pointers compile on the host pass too
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
typedef float fvec4u __attribute__((ext_vector_type(4), aligned(4))); // four / two neighbouring texels, dword aligned
typedef float fvec2u __attribute__((ext_vector_type(2), aligned(4)));
#define DSM_GLOBAL __attribute__((address_space(1)))

// ------------------------------------------------------------------------------------------
// Lane masks.  What a vector instruction costs on gfx950 depends on its FORM (tools/microbench/valu_issue_cost.hip,
// profiles/r05_valu_issue_cost_w*.json; cycles a wave64 instruction occupies its SIMD): v_mul / v_add / v_sub / v_fmac / v_and /
// v_mov / v_add_u32 in the 4-byte encoding with VGPR or inline-constant sources 2.5; their 8-byte encodings and v_fma_f32 3.1;
// ANYTHING with an SGPR source, every v_cmp, v_cndmask, v_cvt, v_fract, v_min / v_max, shifts, integer multiplies 4.65; v_rcp 8.5.
// The level-0 loop is bound by exactly that sum (a padding experiment follows it to 1 %: profiles/r05_ab_pad_level0.log), and
// the compiler's handling of C++ bools was a fifth of it: a bool that crosses a loop edge or feeds a ballot is materialised as
// v_cndmask 0/1 + v_cmp_ne (two 4.65-cycle instructions per ballot), `cond ? x : 0` is a 4.65-cycle select per value.  The
// per-point code therefore keeps its predicates as explicit 64-bit lane masks in SGPR pairs -- a compare writes one (its only
// cost), logic and population counts on them are scalar instructions -- turns a mask into all-ones / zero lane bits ONCE
// (one v_cndmask) and masks values with v_and (2.5 cycles).
// ------------------------------------------------------------------------------------------
typedef unsigned long long lmask;
// LLVM's compare predicate codes (llvm.amdgcn.fcmp / icmp)
enum { kOEQ = 1, kOGT = 2, kOGE = 3, kOLT = 4, kSLT = 40 };
#define DSM_FCMP(a, b, pred) ((lmask)__builtin_amdgcn_fcmpf((a), (b), (pred)))
__device__ __forceinline__ lmask lanes_finite(float x) { return DSM_FCMP(__builtin_fabsf(x), __builtin_inff(), kOLT); } // false for NaN
// A lane mask is the same in every lane by construction; where the compiler cannot see that (a mask chosen under a condition it takes
// for divergent, e.g. the thread group's chunk in coarse_kernel) this pins it to an SGPR pair -- no instruction where it already is one.
__device__ __forceinline__ lmask mask_sgpr(lmask m) {
  return ((lmask)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
// bit set -> t, clear -> f  (v_cndmask_b32 with the mask in an SGPR pair)
__device__ __forceinline__ float sel_f(lmask m, float f, float t) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask_sgpr(m)));
  return r;
}
__device__ __forceinline__ unsigned sel_u(lmask m, unsigned f, unsigned t) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask_sgpr(m)));
  return r;
}
// all-ones where the lane's bit is set, zero elsewhere (opaque to the optimiser on purpose: `x & lane_bits(m)` stays a v_and)
__device__ __forceinline__ unsigned lane_bits(lmask m) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(r) : "s"(mask_sgpr(m)));
  return r;
}
__device__ __forceinline__ float and_f(float x, unsigned bits) { return __uint_as_float(__float_as_uint(x) & bits); }

// getInterpolatedElement33 (upstream DSO), call sites TrackerAndScaler.cpp:790,1106, split in two
// so that the tap loads of one point can be in flight while the previous point is consumed.
//
// The target lives in HBM as its intensities only (4 bytes per texel).  The reference's texel is
// (I, dx, dy) with dx = 0.5 (I[x+1] - I[x-1]), dy = 0.5 (I[y+1] - I[y-1]) (upstream DSO FrameHessian::makeImages), so the
// interpolated gradients follow from a 4 x 4 neighbourhood of intensities of which only 12 values are needed:
//   rows y, y+1: columns x-1 .. x+2 (two 16-byte loads),  rows y-1, y+2: columns x, x+1 (two 8-byte loads)
// -- as many vector-memory instructions and registers as four 12-byte texels, one third of the image bytes.
struct Taps {
  float r1[4], r2[4]; // rows y, y+1: columns x-1 .. x+2
  float r0[2], r3[2]; // rows y-1, y+2: columns x, x+1
  float dx, dy;       // fractional position
};

// The four row bases of a target plane (wave-uniform, kept in SGPRs): every tap load of a point then shares ONE 32-bit
// vector offset (global_load with saddr + voffset, no 64-bit vector address arithmetic).
struct TapBases {
  const DSM_GLOBAL char *r0, *r1, *r2, *r3;
};
__device__ __forceinline__ TapBases tap_bases(const DSM_GLOBAL float *img, int w) {
  const DSM_GLOBAL char *cb = (const DSM_GLOBAL char *)img;
  const long pitch = 4l * w;
  return TapBases{cb - pitch, cb - 4, cb + pitch - 4, cb + 2 * pitch};
}
// The same four row bases for a plane staged in LDS (coarse_kernel): byte addresses in the local address space; the taps
// become ds_read instructions (unaligned rows: the compiler picks ds_read2_b32 / ds_read_b64 as alignment allows).
#define DSM_LDS __attribute__((address_space(3)))
struct TapBasesL {
  unsigned r0, r1, r2, r3;
};
__device__ __forceinline__ TapBasesL tap_bases_lds(unsigned img, int w) {
  const unsigned pitch = 4u * (unsigned)w;
  return TapBasesL{img - pitch, img - 4u, img + pitch - 4u, img + 2u * pitch};
}
// byte offset of the taps of position (x, y): texel (floor x, floor y), or the safe texel (2, 2) for lanes outside `ok` -- ONE
// select, on the finished offset (an unusable lane's x / y may be anything, NaN included: its fractions and values are masked later)
__device__ __forceinline__ unsigned tap_offset(float x, float y, int w, lmask ok, unsigned safe_off, float &dx, float &dy) {
  const int ix = (int)x, iy = (int)y;
  // x - (float)(int)x for x >= 0 is exact and equals v_fract_f32(x) (= x - floor(x))
  dx = __builtin_amdgcn_fractf(x);
  dy = __builtin_amdgcn_fractf(y);
  const unsigned off = 4u * ((unsigned)ix + (unsigned)iy * (unsigned)w);
  return sel_u(ok, safe_off, off);
}
template <bool L> struct TapSel { typedef TapBases type; };
template <> struct TapSel<true> { typedef TapBasesL type; };
template <bool GRAD = true>
__device__ __forceinline__ void taps_load(const TapBasesL &B, float x, float y, int w, lmask ok, unsigned safe_off, Taps &T) {
  const unsigned off = tap_offset(x, y, w, ok, safe_off, T.dx, T.dy);
  if (GRAD) {
    const fvec4u a = *(const DSM_LDS fvec4u *)(size_t)(B.r1 + off);
    const fvec4u b = *(const DSM_LDS fvec4u *)(size_t)(B.r2 + off);
    const fvec2u c = *(const DSM_LDS fvec2u *)(size_t)(B.r0 + off);
    const fvec2u d = *(const DSM_LDS fvec2u *)(size_t)(B.r3 + off);
    T.r1[0] = a.x, T.r1[1] = a.y, T.r1[2] = a.z, T.r1[3] = a.w;
    T.r2[0] = b.x, T.r2[1] = b.y, T.r2[2] = b.z, T.r2[3] = b.w;
    T.r0[0] = c.x, T.r0[1] = c.y;
    T.r3[0] = d.x, T.r3[1] = d.y;
  } else {
    const fvec2u a = *(const DSM_LDS fvec2u *)(size_t)(B.r1 + off + 4u);
    const fvec2u b = *(const DSM_LDS fvec2u *)(size_t)(B.r2 + off + 4u);
    T.r1[0] = T.r1[3] = T.r2[0] = T.r2[3] = 0.f;
    T.r1[1] = a.x, T.r1[2] = a.y;
    T.r2[1] = b.x, T.r2[2] = b.y;
    T.r0[0] = T.r0[1] = T.r3[0] = T.r3[1] = 0.f;
  }
}
// GRAD = false (residual-only evaluations): the intensity needs columns x, x+1 of rows y, y+1 only -- two 8-byte loads.
template <bool GRAD = true>
__device__ __forceinline__ void taps_load(const TapBases &B, float x, float y, int w, lmask ok, unsigned safe_off, Taps &T) {
  const unsigned off = tap_offset(x, y, w, ok, safe_off, T.dx, T.dy);
  if (GRAD) {
    const fvec4u a = *(const DSM_GLOBAL fvec4u *)(B.r1 + off); // (x-1 .. x+2, y)
    const fvec4u b = *(const DSM_GLOBAL fvec4u *)(B.r2 + off); // (x-1 .. x+2, y+1)
    const fvec2u c = *(const DSM_GLOBAL fvec2u *)(B.r0 + off); // (x, x+1; y-1)
    const fvec2u d = *(const DSM_GLOBAL fvec2u *)(B.r3 + off); // (x, x+1; y+2)
    T.r1[0] = a.x, T.r1[1] = a.y, T.r1[2] = a.z, T.r1[3] = a.w;
    T.r2[0] = b.x, T.r2[1] = b.y, T.r2[2] = b.z, T.r2[3] = b.w;
    T.r0[0] = c.x, T.r0[1] = c.y;
    T.r3[0] = d.x, T.r3[1] = d.y;
  } else { // the intensity alone: (x, x+1) of rows y and y+1
    const fvec2u a = *(const DSM_GLOBAL fvec2u *)(B.r1 + off + 4u);
    const fvec2u b = *(const DSM_GLOBAL fvec2u *)(B.r2 + off + 4u);
    T.r1[0] = T.r1[3] = T.r2[0] = T.r2[3] = 0.f;
    T.r1[1] = a.x, T.r1[2] = a.y;
    T.r2[1] = b.x, T.r2[2] = b.y;
    T.r0[0] = T.r0[1] = T.r3[0] = T.r3[1] = 0.f;
  }
}

// h0 (the intensity) decides in/out, Huber and cut-off: exact reference operation order.  g1 / g2 are TWICE the
// interpolated gradients (the 0.5 of the central difference is a power of two: it commutes with every rounding of the
// chain and is folded into the focal length by the caller); they only enter the Jacobian sums, which are compared to
// float tolerance: FMA allowed.  makeImages replaces a non-finite gradient by zero; some tap difference is non-finite
// exactly when the interpolated value is (the weights are finite), which the caller tests per wave: EXACT = the rare
// path that applies the replacement tap by tap.
template <bool EXACT>
__device__ __forceinline__ void taps_gradients(const Taps &T, float w00, float w10, float w01, float w11, float &g1, float &g2) {
  auto fix = [](float d) { return EXACT ? (__builtin_isfinite(d) ? d : 0.0f) : d; };
  const float d00 = fix(T.r1[2] - T.r1[0]), d10 = fix(T.r1[3] - T.r1[1]);
  const float d01 = fix(T.r2[2] - T.r2[0]), d11 = fix(T.r2[3] - T.r2[1]);
  const float e00 = fix(T.r2[1] - T.r0[0]), e10 = fix(T.r2[2] - T.r0[1]);
  const float e01 = fix(T.r3[0] - T.r1[1]), e11 = fix(T.r3[1] - T.r1[2]);
  g1 = __builtin_fmaf(w00, d00, __builtin_fmaf(w10, d10, __builtin_fmaf(w01, d01, w11 * d11)));
  g2 = __builtin_fmaf(w00, e00, __builtin_fmaf(w10, e10, __builtin_fmaf(w01, e01, w11 * e11)));
}
template <bool EXACT, bool GRAD = true>
__device__ __forceinline__ void taps_interp(const Taps &T, float &h0, float &g1, float &g2) {
  const float dx = T.dx, dy = T.dy;
  const float dxdy = dx * dy;
  const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  h0 = ((w11 * T.r2[2] + w01 * T.r2[1]) + w10 * T.r1[2]) + w00 * T.r1[1];
  if (GRAD)
    taps_gradients<EXACT>(T, w00, w10, w01, w11, g1, g2);
  else
    g1 = g2 = 0.f;
}

// per-point state carried from the warp stage to the consume stage
struct Warped {
  float u, v, new_idepth, refColor;
  float x, y, id; // scale mode only (rx = M (x,y,1) / id, :1068)
  lmask inb;      // lanes whose point is usable (:786 / :1102): wave-uniform, lives in an SGPR pair
};

// ------------------------------------------------------------------------------------------
// eval kernel
// ------------------------------------------------------------------------------------------
// Wave-uniform inputs of one evaluation (kept in SGPRs): a copy of EvalIn.
struct EvalConsts {
  const float4 *pts;
  const float *img;
  int n, w, h;
  int ppt; // EvalIn::ppt
  float fx, fy, cx, cy;
  float Ki[9];
  float huber;
  float M[9];
  float t[3];
  float aff0, aff1, b0, scale, cutoff, max_energy;
  int residual_only; // EvalIn::residual_only
  unsigned lds_img, lds_pts; // coarse_kernel: LDS byte addresses of the staged intensity plane / template (else unused)
};
__device__ __forceinline__ void eval_consts_defaults(EvalConsts &c) { c.lds_img = c.lds_pts = 0; }

// One chunk of one evaluation by 256 threads (tid = 0..255 inside the chunk's thread group):
// the per-point loop, the flow-indicator pass and the fixed-order reduction into the chunk's 52-slot
// partial `out` (global memory in eval_kernel, LDS in coarse_kernel).  `red` is this thread group's
// [16][kNumSlots] LDS scratch.  Contains two workgroup barriers: every thread of the workgroup must
// call it; thread groups without a chunk pass active = false.
// RO = residual-only (EvalIn::residual_only): the residual side alone -- energy, counts, flow indicators; no gradients, no
// Jacobian, no normal-equation sums (their partial slots are written as zeros), two tap loads per point instead of four.
// LDSIMG / LDSPTS: the target plane / the template are read from their LDS copies (c.lds_img / c.lds_pts) -- same values,
// same operations, hence the same results as from global memory.
// arrive != nullptr (eval_kernel without the fused LM step): instead of a workgroup barrier between the row sums and the
// final sum, every wave takes a ticket on an LDS counter after its rows are written and only the LAST one stays to add the 16
// rows and store the partial -- the same additions in the same order by another wave; the other waves leave at once instead
// of waiting for the slowest wave's gathers (measured: a mid-level workgroup spent a third of its life between the end of
// its first wave's loop and the partial store).
// DEEP: the two-points-per-trip loop with fixed register roles: level 0 everywhere, and every level inside the tick engine's kernel,
// which runs at four waves per SIMD whatever the level.  (Round 5 measured it there on the 16-point chunks of the other levels and saw
// no change, profiles/r05_ab_flow_first_deep16_b1.log; round 6's shader-clock stamps found the one-point loop's items of levels 1-2
// resident 13 % longer than level 0's for the same 4096 points -- 34 % for the residual-only ones, which have two gathers and little
// arithmetic between them -- and the same A/B now gives + 1.3 % frames/s, the kernel 0.61 -> 0.63 of peak: profiles/r06_ab_deep_all_levels.log)
// VC: which wave-uniform constants the loop keeps in VGPRs (an SGPR source costs an instruction 4.65 instead of 2.5 cycles): 0 none
// (the 96-register kernels of five waves per SIMD), 1 the warp's twelve (the two-point loop at four waves per SIMD: 128 registers hold
// these and no more), 2 the camera, gradient-scale and brightness constants as well (the one-point loop inside a kernel that is
// allocated 128 registers anyway: tick_eval_kernel)
template <int MODE, bool LVL0, bool RO, bool LDSIMG = false, bool LDSPTS = false, bool DEEP = LVL0, int VC = DEEP ? 1 : 0>
__device__ __forceinline__ void eval_chunk_impl(const EvalConsts &c, int chunk, int tid, bool active, float (*red)[kNumSlots],
                                                float *out, int *arrive = nullptr) {
  const int n = c.n;
  const int P = c.ppt;
  constexpr int NACC = MODE == 1 ? 3 : kNumAcc;
  float acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; a++) acc[a] = 0.f;
  float E = 0.f;
  int n_terms = 0, n_sat = 0, n_warped = 0;
  float fT = 0.f, fRT = 0.f, fNum = 0.f;
  if (active) {
    const float M0 = c.M[0], M1 = c.M[1], M2 = c.M[2], M3 = c.M[3], M4 = c.M[4], M5 = c.M[5],
                M6 = c.M[6], M7 = c.M[7], M8 = c.M[8];
    const float t0 = c.t[0], t1 = c.t[1], t2 = c.t[2];
    const float cutoff = c.cutoff;
    const float huber = c.huber;
    const float fxl = c.fx, fyl = c.fy, cxl = c.cx, cyl = c.cy;
    const float hfx = 0.5f * fxl, hfy = 0.5f * fyl; // the central differences' 0.5 (see taps_interp)
    const int wl = c.w, hl = c.h;
    const float wm3 = (float)(wl - 3), hm3 = (float)(hl - 3);
    const int pitch = wl; // floats per row of what the taps are fetched from
    typename TapSel<LDSIMG>::type img;
    if constexpr (LDSIMG)
      img = tap_bases_lds(c.lds_img, pitch);
    else
      img = tap_bases((const DSM_GLOBAL float *)c.img, wl);
    const TapBases img_g = tap_bases((const DSM_GLOBAL float *)c.img, wl); // the plane in HBM (the rare non-finite-gradient path re-fetches from it)
    const DSM_GLOBAL fvec4 *pts = (const DSM_GLOBAL fvec4 *)c.pts;
    // scale mode: (scale * M) is formed once per evaluation, as `scale * rot_f1_f0_K0_i` is (:1061)
    const float sc = c.scale;
    const float S0 = sc * M0, S1 = sc * M1, S2 = sc * M2, S3 = sc * M3, S4 = sc * M4, S5 = sc * M5,
                S6 = sc * M6, S7 = sc * M7, S8 = sc * M8;
    const float aff0 = c.aff0, aff1 = c.aff1, b0 = c.b0;

    const int chunk_start = chunk * kThreads * P;

    // Software-pipelined, branch-free loop.  Stage A warps template point k+1 and issues its four
    // bilinear tap loads; stage B consumes the taps of point k (residual, Huber, Jacobian, 45 FMAs)
    // while those loads are in flight.  Predicates are lane masks (see "Lane masks" above): unusable lanes fetch from a safe
    // address and their values are ANDed to zero, so the accumulators never cross a divergent join.
    const lmask full = __builtin_amdgcn_ballot_w64(true);
    unsigned safe_off = 4u * (2u + 2u * (unsigned)pitch); // byte offset of texel (2, 2): held in a VGPR (a select takes no SGPR besides its mask)
    asm volatile("" : "+v"(safe_off));
    // The warp's twelve constants in VGPRs: an SGPR source costs an instruction 4.65 instead of 2.5 cycles (level-0 loop: 1163 -> 1111
    // cycles per two points).  The register budget of four waves per SIMD (128) has room for these and no more.
    float vM0 = M0, vM1 = M1, vM2 = M2, vM3 = M3, vM4 = M4, vM5 = M5, vM6 = M6, vM7 = M7, vM8 = M8, vt0 = t0, vt1 = t1, vt2 = t2;
    if (VC >= 1 && MODE == 0) asm volatile("" : "+v"(vM0), "+v"(vM1), "+v"(vM2), "+v"(vM3), "+v"(vM4), "+v"(vM5), "+v"(vM6), "+v"(vM7), "+v"(vM8), "+v"(vt0), "+v"(vt1), "+v"(vt2));
    float vfx = fxl, vfy = fyl, vcx = cxl, vcy = cyl, vhfx = hfx, vhfy = hfy, vaff0 = aff0, vaff1 = aff1, vb0 = b0;
    if (VC >= 2 && MODE == 0) asm volatile("" : "+v"(vfx), "+v"(vfy), "+v"(vcx), "+v"(vcy), "+v"(vhfx), "+v"(vhfy), "+v"(vaff0), "+v"(vaff1), "+v"(vb0));
    auto stage_a = [&](const fvec4 &p, lmask in_list, Warped &W, Taps &T) {
      const float x = p.x, y = p.y, id = p.z;
      float pt0, pt1, pt2;
      if (MODE == 0) { // :747
        pt0 = ((vM0 * x + vM1 * y) + vM2) + vt0 * id;
        pt1 = ((vM3 * x + vM4 * y) + vM5) + vt1 * id;
        pt2 = ((vM6 * x + vM7 * y) + vM8) + vt2 * id;
      } else if (MODE == 2) { // PoseEstimator.cpp:192: pt = R (x,y,z) + t ; the float4 is (x, y, z, refColor[lvl])
        pt0 = ((M0 * x + M1 * y) + M2 * id) + t0;
        pt1 = ((M3 * x + M4 * y) + M5 * id) + t1;
        pt2 = ((M6 * x + M7 * y) + M8 * id) + t2;
      } else { // :1061
        pt0 = ((S0 * x + S1 * y) + S2) + t0 * id;
        pt1 = ((S3 * x + S4 * y) + S5) + t1 * id;
        pt2 = ((S6 * x + S7 * y) + S8) + t2 * id;
      }
      // u = pt0/pt2, v = pt1/pt2, new_idepth = id/pt2 (:748-752).  u and v position the bilinear
      // taps and decide in/out, so they must equal the IEEE quotients bit for bit; new_idepth only
      // enters the sign test and the Jacobian.  The three quotients share one refined reciprocal
      // and use the hardware's own correction sequence (what `a / d` expands to) without the
      // range scaling / fix-up instructions, which are identities for ordinary operands: 2^-33 <= |pt2| < 2^32 and
      // |id| >= 2^-64 or id = 0.  A wave that holds a lane outside that range takes the full IEEE divisions instead.
      {
        const float apt2 = __builtin_fabsf(pt2);
        lmask ordinary = DSM_FCMP(apt2, 0x1p-33f, kOGE) & DSM_FCMP(apt2, 0x1p32f, kOLT);
        if (MODE != 2) ordinary &= DSM_FCMP(__builtin_fabsf(id), 0x1p-64f, kOGE) | DSM_FCMP(id, 0.0f, kOEQ);
        if (__builtin_expect(ordinary != full, 0)) {
          W.u = pt0 / pt2;
          W.v = pt1 / pt2;
          W.new_idepth = (MODE == 2 ? 1.0f : id) / pt2; // PoseEstimator.cpp:197: 1 / pt[2]
        } else {
          const float r0 = __builtin_amdgcn_rcpf(pt2);
          const float r1 = __builtin_fmaf(__builtin_fmaf(-pt2, r0, 1.0f), r0, r0);
          auto quot = [pt2, r1](float a) {
            const float q0 = a * r1;
            const float q1 = __builtin_fmaf(__builtin_fmaf(-pt2, q0, a), r1, q0);
            return __builtin_fmaf(__builtin_fmaf(-pt2, q1, a), r1, q1);
          };
          W.u = quot(pt0);
          W.v = quot(pt1);
          W.new_idepth = MODE == 2 ? r1 : id * r1;
        }
      }
      const float Ku = vfx * W.u + vcx, Kv = vfy * W.v + vcy;
      W.refColor = p.w;
      W.x = x, W.y = y, W.id = id;
      // :786 / :1102 (every comparison is false for a NaN, as in the reference)
      W.inb = in_list & DSM_FCMP(Ku, 2.0f, kOGT) & DSM_FCMP(Kv, 2.0f, kOGT) & DSM_FCMP(Ku, wm3, kOLT) & DSM_FCMP(Kv, hm3, kOLT) &
              DSM_FCMP(W.new_idepth, 0.0f, kOGT);
      taps_load<!RO>(img, Ku, Kv, pitch, W.inb, safe_off, T); // (unusable lanes fetch around texel (2, 2))
    };
    auto stage_b = [&](const Warped &W, const Taps &T) {
      float h0, g1, g2;
      taps_interp<false, !RO>(T, h0, g1, g2);
      // makeImages' "non-finite gradient -> 0", the rare path: the taps are fetched again (so that the common path
      // does not keep twelve registers alive for it) and the replacement is applied tap by tap
      if (!RO && __builtin_expect((W.inb & ~(lanes_finite(g1) & lanes_finite(g2))) != 0ull, 0)) {
        Taps Tx; // (always from the plane in HBM: the same values as a staged copy holds)
        taps_load(img_g, fxl * W.u + cxl, fyl * W.v + cyl, wl, W.inb, safe_off, Tx);
        taps_interp<true>(Tx, h0, g1, g2);
      }
      const float refColor = W.refColor;
      const lmask fin = W.inb & lanes_finite(h0); // :791
      const float residual = MODE != 1 ? h0 - (vaff0 * refColor + vaff1) : h0 - refColor; // :793 / :1109
      const float ar = __builtin_fabsf(residual);
      // Huber weight (:794-795): 1 for |r| < huber, huber / |r| beyond.  It is off the per-point decision path (in/out, cut-off):
      // it only scales this point's terms of E and of the normal equations -- sums that are compared to tolerance, E then
      // entering the LM accept test like any other rounding of the sum -- so the hardware reciprocal (1 ulp) replaces the IEEE
      // division and the branch is a minimum (huber / |r| > 1 exactly where |r| < huber, up to that ulp; 1 for r = 0).
      const float hw = __builtin_fminf(1.0f, huber * __builtin_amdgcn_rcpf(ar));
      const lmask sat = DSM_FCMP(ar, cutoff, kOGT); // :797
      const lmask use = fin & ~sat;
      // integer outputs are counted per wave on the scalar unit (s_bcnt1 of the lane masks)
      n_terms += __builtin_popcountll(fin);
      n_sat += __builtin_popcountll(fin & sat);
      n_warped += __builtin_popcountll(use);
      const unsigned m = lane_bits(use);
      // :809 for the usable lanes; a saturated lane's term is the constant max_energy (:800): n_sat x max_energy joins E where the
      // partials are reduced (build_rs)
      E += and_f(hw * residual * residual * (2 - hw), m);
      const float wgt = and_f(hw, m);
      if (RO) {
        // nothing else: the sums below feed calcGSSSE*, whose output the ending loop never reads
      } else if (MODE != 1) {
        // calcGSSSEPose :658-678 on the values calcResPose would have buffered (:812-819); masked
        // lanes get all-zero inputs so that they add exact zeros
        const float u = and_f(W.u, m), v = and_f(W.v, m), nid = and_f(W.new_idepth, m);
        const float dx = and_f(g1 * vhfx, m), dy = and_f(g2 * vhfy, m); // dxInterp = hitColor[1] * fx (:814), 0.5 folded
        float J[9];
        J[0] = nid * dx;
        J[1] = nid * dy;
        J[2] = -(nid * __builtin_fmaf(u, dx, v * dy));
        J[3] = -__builtin_fmaf(u * v, dx, dy * __builtin_fmaf(v, v, 1.0f));
        J[4] = __builtin_fmaf(u * v, dy, dx * __builtin_fmaf(u, u, 1.0f));
        J[5] = __builtin_fmaf(u, dy, -(v * dx));
        J[6] = and_f(vaff0 * (vb0 - refColor), m);
        J[7] = -1.0f;
        J[8] = and_f(residual, m);
        int idx = 0;
  #pragma unroll
        for (int r = 0; r < 9; r++) { // Accumulator9::updateSSE_eighted: H(r,c) += (J_r w) J_c
          const float Jw = J[r] * wgt;
  #pragma unroll
          for (int c = r; c < 9; c++) {
            acc[idx] = __builtin_fmaf(Jw, J[c], acc[idx]);
            idx++;
          }
        }
      } else {
        // calcResScale :1068 and calcGSSSEScale :983-999
        const float x = W.x, y = W.y, id = W.id;
        const float rx1 = ((M0 * x + M1 * y) + M2) / id;
        const float rx2 = ((M3 * x + M4 * y) + M5) / id;
        const float rx3 = ((M6 * x + M7 * y) + M8) / id;
        const float dxfx = g1 * hfx, dyfy = g2 * hfy;
        const float deno_sqrt = sc * rx3 + t2;
        const float deno = 1.0f / (deno_sqrt * deno_sqrt);
        const float xno = rx1 * t2 - rx3 * t0;
        const float yno = rx2 * t2 - rx3 * t1;
        const float J0 = and_f(dxfx * (deno * xno) + dyfy * (deno * yno), m);
        const float J1 = and_f(residual, m);
        const float J0w = J0 * wgt;
        acc[0] = __builtin_fmaf(J0w, J0, acc[0]);
        acc[1] = __builtin_fmaf(J0w, J1, acc[1]);
        acc[2] = __builtin_fmaf(J1 * wgt, J1, acc[2]);
      }
    };

    // flow indicators (:754-784 / :1070-1100): level 0, every 32nd template index.  Two waves handle the 8*P such points of this chunk
    // in a single pass.  In the two-point loop's kernels (FLOW_FIRST) the pass runs BEFORE the loop, its point load travelling with the
    // first template entries the workgroup has to wait for anyway, and its three row sums wait in the reduction scratch instead of
    // in registers: after the loop the pass was a dependent HBM round trip plus four divisions in front of the reduction -- 5 k of a
    // level-0 workgroup's 65 k cycles (shader-clock stamps, profiles/r05_queue_probe_and_geometry.log); same operations on the same
    // values, same row sums: the same bits.  Measured: tick_eval_kernel + 2.4 %, stream + 1.4 % (profiles/r05_ab_flow_first.log).
    constexpr bool FLOW_FIRST = LVL0 && DEEP;
    auto flow_pass = [&]() {
    if (LVL0 && tid < 8 * P) {
      const int i = chunk_start + 32 * tid;
      if (i < n) {
        const fvec4 p = pts[i];
        const float x = p.x, y = p.y, id = p.z;
        const float *Ki = c.Ki;
        if (MODE == 2) {
          // PoseEstimator.cpp:185-229, as written: shifts are measured against the reference
          // projection (Ku0,Kv0) and the "translation only" points use (x, y, 1)
          const float z = id;
          const float Ku0 = fxl * (x / z) + cxl, Kv0 = fyl * (y / z) + cyl;
          const float ptz = ((M6 * x + M7 * y) + M8 * z) + t2;
          const float Ku = fxl * ((((M0 * x + M1 * y) + M2 * z) + t0) / ptz) + cxl;
          const float Kv = fyl * ((((M3 * x + M4 * y) + M5 * z) + t1) / ptz) + cyl;
          const float KuT = fxl * ((x + t0) / (1.0f + t2)) + cxl, KvT = fyl * ((y + t1) / (1.0f + t2)) + cyl;
          const float KuT2 = fxl * ((x - t0) / (1.0f - t2)) + cxl, KvT2 = fyl * ((y - t1) / (1.0f - t2)) + cyl;
          const float p3z = ((M6 * x + M7 * y) + M8) - t2;
          const float Ku3 = fxl * ((((M0 * x + M1 * y) + M2) - t0) / p3z) + cxl;
          const float Kv3 = fyl * ((((M3 * x + M4 * y) + M5) - t1) / p3z) + cyl;
          fT += (KuT - Ku0) * (KuT - Ku0) + (KvT - Kv0) * (KvT - Kv0);
          fT += (KuT2 - Ku0) * (KuT2 - Ku0) + (KvT2 - Kv0) * (KvT2 - Kv0);
          fRT += (Ku - Ku0) * (Ku - Ku0) + (Kv - Kv0) * (Kv - Kv0);
          fRT += (Ku3 - Ku0) * (Ku3 - Ku0) + (Kv3 - Kv0) * (Kv3 - Kv0);
        } else {
          float kx0, kx1, kx2, rx0, rx1, rx2;
          if (MODE == 0) {
            kx0 = (Ki[0] * x + Ki[1] * y) + Ki[2];
            kx1 = (Ki[3] * x + Ki[4] * y) + Ki[5];
            kx2 = (Ki[6] * x + Ki[7] * y) + Ki[8];
            rx0 = (M0 * x + M1 * y) + M2;
            rx1 = (M3 * x + M4 * y) + M5;
            rx2 = (M6 * x + M7 * y) + M8;
          } else {
            kx0 = ((sc * Ki[0]) * x + (sc * Ki[1]) * y) + (sc * Ki[2]);
            kx1 = ((sc * Ki[3]) * x + (sc * Ki[4]) * y) + (sc * Ki[5]);
            kx2 = ((sc * Ki[6]) * x + (sc * Ki[7]) * y) + (sc * Ki[8]);
            rx0 = (S0 * x + S1 * y) + S2;
            rx1 = (S3 * x + S4 * y) + S5;
            rx2 = (S6 * x + S7 * y) + S8;
          }
          const float a0 = t0 * id, a1 = t1 * id, a2 = t2 * id;
          const float ptz = rx2 + a2;
          const float Ku = fxl * ((rx0 + a0) / ptz) + cxl, Kv = fyl * ((rx1 + a1) / ptz) + cyl;
          const float pTz = kx2 + a2;
          const float KuT = fxl * ((kx0 + a0) / pTz) + cxl, KvT = fyl * ((kx1 + a1) / pTz) + cyl;
          const float pT2z = kx2 - a2;
          const float KuT2 = fxl * ((kx0 - a0) / pT2z) + cxl, KvT2 = fyl * ((kx1 - a1) / pT2z) + cyl;
          const float p3z = rx2 - a2;
          const float Ku3 = fxl * ((rx0 - a0) / p3z) + cxl, Kv3 = fyl * ((rx1 - a1) / p3z) + cyl;
          fT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
          fT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
          fRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
          fRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
        }
        fNum += 2;
      }
    }
    };
    {
      // Template stream: entry (start + tid) of the list = one coalesced 16-byte load per lane from a wave-uniform base (SGPR pair)
      // plus this thread's constant byte offset -- no per-point index arithmetic.  Entries past the chunk (the loop prefetches up to
      // three trips ahead) or past the list are never used: their lanes are masked, and an entry past the list reads as zeros (the
      // buffer descriptor checks the range); the LDS copy (coarse_kernel) is read with a clamped index instead.
      const unsigned voff = 16u * (unsigned)tid;
      const unsigned lds_pts = c.lds_pts;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)c.pts, 0, 16 * n, 0x00020000);
      auto load_pt = [=](int start) {
        if constexpr (LDSPTS) {
          const int idx = start + tid;
          return *(const DSM_LDS fvec4 *)(size_t)(lds_pts + 16u * (unsigned)(idx < n ? idx : n - 1));
        } else {
          // streamed once: non-temporal (aux = 2), so the template does not evict target rows from the 32 KiB L1 (+2.5 %).
          // buffer_load ... offen: descriptor and the trip's byte offset in SGPRs, this thread's constant offset in a VGPR -- no vector
          // address arithmetic at all; an entry beyond the list reads as zeros (range-checked by the descriptor).
          const uvec4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16 * start, 2);
          return fvec4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
      };
      // lanes whose entry `start + tid` exists and whose trip belongs to the chunk: every lane except in the list's last chunk
      const bool tail = chunk_start + kThreads * P > n;
      auto listed = [=](int start, bool trip) -> lmask {
        if (!trip) return 0ull;
        return mask_sgpr(tail ? (lmask)__builtin_amdgcn_sicmp(tid, n - start, kSLT) : full);
      };
      const int i = chunk_start;
      const fvec4 p0 = load_pt(i);
      if (DEEP) {
        // Level 0 (long loops, HBM-resident targets): two points per trip with FIXED register roles (sets a / b).
        // The taps of point k+1 are issued before the arithmetic of point k and awaited only after the taps of
        // point k+2 have been issued, so two points' gathers are in flight per wave; template entries are fetched
        // two points ahead.  (The rotating single-body loop below makes the compiler copy the freshly loaded tap
        // registers at the back-edge, which waits for them -- vmcnt(0) -- and leaves only the arithmetic of one
        // point to cover the memory latency; it needs 7 VGPRs less, which is one more resident wave per SIMD and
        // worth more than the deeper pipeline on the small, cache-resident levels: level 0 +5 %, levels 1-3 -7..-10 %
        // with this form.)  Points beyond the chunk / the list are masked (they add exact zeros), so odd P needs
        // no special case.
        fvec4 ea = load_pt(i + kThreads), eb = load_pt(i + 2 * kThreads);
        if (FLOW_FIRST) {
          flow_pass();
          const float sT = row16_sum(fT), sRT = row16_sum(fRT), sN = row16_sum(fNum);
          if ((tid & 15) == 0) {
            const int row_ = (tid >> 6) * 4 + ((tid & 63) >> 4);
            red[row_][kSlotFlowT] = sT, red[row_][kSlotFlowRT] = sRT, red[row_][kSlotFlowNum] = sN;
          }
          fT = fRT = fNum = 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0); // drain the prologue loads so the loop's waits only cover its own loads
        Warped Wa, Wb;
        Taps Ta, Tb;
        stage_a(p0, listed(i, true), Wa, Ta);
        int ia = i + kThreads; // start of the entries held in ea (eb: ia + kThreads)
        for (int k = 0; k < P; k += 2) {
          stage_a(ea, listed(ia, k + 1 < P), Wb, Tb); // point k+1
          ea = load_pt(ia + 2 * kThreads);
          stage_b(Wa, Ta); // point k
          stage_a(eb, listed(ia + kThreads, k + 2 < P), Wa, Ta); // point k+2
          eb = load_pt(ia + 3 * kThreads);
          stage_b(Wb, Tb); // point k+1 (masked when k+1 == P)
          ia += 2 * kThreads;
        }
      } else {
        // Template stream prefetched one point ahead.
        int i2 = i + kThreads;
        fvec4 p_next = load_pt(i2);
        __builtin_amdgcn_s_waitcnt(0); // drain the prologue loads so the loop's waits only cover its own loads
        Warped Wc;
        Taps Tc;
        stage_a(p0, listed(i, true), Wc, Tc);
        for (int k = 0; k < P; k++) {
          // stage A for point k+1 (its template entry was prefetched one iteration ago)
          const fvec4 p = p_next;
          const lmask in_next = listed(i2, k + 1 < P);
          const int i3 = i2 + kThreads;
          p_next = load_pt(i3);
          Warped Wn;
          Taps Tn;
          stage_a(p, in_next, Wn, Tn);
          // stage B for point k
          stage_b(Wc, Tc);
          Wc = Wn;
          Tc = Tn;
          i2 = i3;
        }
      }
    }

    if (!FLOW_FIRST) flow_pass();
  } // active
  // ---- workgroup reduction: DPP row sums -> LDS [16 rows][slots] -> fixed-order sum ----
  const int lane = tid & 63, wave = tid >> 6;
  const int row = wave * 4 + (lane >> 4);
  const bool writer = (lane & 15) == 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) {
    const float s = RO ? 0.0f : row16_sum(acc[i]);
    if (writer) red[row][i] = s;
  }
  {
    // flow indicators: level 0 only; `parked`: this thread group's sums already sit in red[][] (FLOW_FIRST above)
    const bool parked = LVL0 && DEEP && active;
    const bool flow_here = LVL0 && !parked;
    const float sE = row16_sum(E), sT = flow_here ? row16_sum(fT) : 0.f, sRT = flow_here ? row16_sum(fRT) : 0.f, sN = flow_here ? row16_sum(fNum) : 0.f;
    // n_terms / n_sat / n_warped are wave totals (identical in every lane): count them once per wave
    const bool first = lane == 0;
    const int iT = first ? n_terms : 0, iS = first ? n_sat : 0, iW = first ? n_warped : 0; // (a row's writer is its lane 0: the wave's first row carries the totals)
    if (writer && !parked) {
      red[row][kSlotFlowT] = sT;
      red[row][kSlotFlowRT] = sRT;
      red[row][kSlotFlowNum] = sN;
    }
    if (writer) {
      red[row][kSlotE] = sE;
      red[row][kSlotNTerms] = __int_as_float(iT);
      red[row][kSlotNSat] = __int_as_float(iS);
      red[row][kSlotNWarped] = __int_as_float(iW);
    }
  }
  int slot = tid;
  if (arrive) {
    // a wave's LDS operations are performed in order: its rows are in place before its ticket is counted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(t) != kThreads / 64 - 1) return; // not the last wave of the workgroup: done
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    slot = lane;
  } else {
    __syncthreads();
  }
  const bool is_float_slot = slot < NACC || (slot >= kSlotE && slot < kSlotNTerms);
  if (!active) {
  } else if (is_float_slot) {
    float s = red[0][slot];
#pragma unroll
    for (int r = 1; r < 16; r++) s += red[r][slot];
    store_partial(out + slot, s);
  } else if (slot >= kSlotNTerms && slot < kNumSlots) {
    int s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += __float_as_int(red[r][slot]);
    store_partial(out + slot, __int_as_float(s));
  }
}

template <int MODE, bool LVL0, bool DEEP = LVL0, int VC = DEEP ? 1 : 0>
__device__ __forceinline__ void eval_chunk(const EvalConsts &c, int chunk, int tid, bool active, float (*red)[kNumSlots],
                                           float *out, int *arrive = nullptr) {
  if (c.residual_only) // wave-uniform
    eval_chunk_impl<MODE, LVL0, true, false, false, DEEP, VC>(c, chunk, tid, active, red, out, arrive);
  else
    eval_chunk_impl<MODE, LVL0, false, false, false, DEEP, VC>(c, chunk, tid, active, red, out, arrive);
}

// ------------------------------------------------------------------------------------------
// LM state machine.  One workgroup of 256 threads per problem: all four waves reduce the chunk
// partials (fixed order), then wave 0 advances the state machine: the 8x8 normal equations live
// one element per lane (lane = 8*row + col), the pivoted LDLT solve runs across the wave with
// shuffles, lane 0 does the scalar double precision work (SE3 exp, pose composition).
// ------------------------------------------------------------------------------------------
constexpr int kLmThreads = 256;

// The inputs of an evaluation, in parts (round 6 -- an LM step is ONE wave's chain of dependent instructions, every one of them is
// time: shader-clock stamps in profiles/r06_tick_stamps.json had make_eval at 6 k of a step's 22 k cycles):
//   make_eval_level  what depends on the level alone (and, for the scale problem, everything but the scale: R10 K^-1 and tsl_f1_f0 are
//                    constants of the level) -- written ONCE when a level begins, into both candidates' blocks;
//   make_eval_rot    M and t of a proposed pose;  make_eval_aff  affLL of a proposed affine pair (the only exp of the step);
//   make_eval_cutoff cutoff, max_energy, the residual-only mark.
// Every value is formed by the same operations on the same operands as the one-piece form of rounds 1-5: same bits.
// `st`: this lane stores (the callers run on wave-uniform values; lane 0 writes).
__device__ __forceinline__ void make_eval_level(const TrackerDev &T, EvalIn &e, int mode, int lvl) {
  const LevelDev &L = T.lv[lvl];
  float Ki[9];
#pragma unroll
  for (int i = 0; i < 9; i++) Ki[i] = L.Ki[i];
#pragma unroll
  for (int i = 0; i < 9; i++) e.Ki[i] = Ki[i];
  e.pts = L.pts;
  e.img = L.img[mode == 1 ? 1 : 0]; // new left frame (:709) / right frame fh1_ (:1016)
  e.n = L.n;
  e.ppt = pts_per_thread(L.n, T.p.geometry), e.pad0 = e.pad1 = e.pad2 = 0;
  e.w = L.w;
  e.h = L.h;
  if (mode == 1) { // cam-1 intrinsics (:1017-1020)
    e.fx = L.fx1, e.fy = L.fy1, e.cx = L.cx1, e.cy = L.cy1;
  } else {
    e.fx = L.fx, e.fy = L.fy, e.cx = L.cx, e.cy = L.cy;
  }
  e.huber = T.p.huber_th;
  e.b0 = mode == 0 ? (float)T.ref_b : 0.0f; // :646 (pose); the loop-closure estimator's reference has b = 0 (PoseEstimator.cpp:90)
  e.scale = 1.0f;
  if (mode == 1) {
    double Rd[9];
    quat_to_rot(T.T10, Rd);
    float Rf[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
    float M[9];
    mat3f_mul(Rf, Ki, M); // :1022-1023
#pragma unroll
    for (int i = 0; i < 9; i++) e.M[i] = M[i];
    e.t[0] = (float)T.T10[4]; // :1024
    e.t[1] = (float)T.T10[5];
    e.t[2] = (float)T.T10[6];
    e.aff0 = 1.0f;
    e.aff1 = 0.0f;
  }
}
// pose (mode 0): M = R K^-1 (:715), t (:716); loop-closure pose (mode 2, PoseEstimator.cpp:155-163): M = R.  e.Ki holds the level's K^-1.
__device__ __forceinline__ void make_eval_rot(EvalIn &e, int mode, const double pose[7], bool st) {
  double Rd[9];
  quat_to_rot(pose, Rd);
  float Rf[9];
#pragma unroll
  for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
  float M[9];
  if (mode == 2) {
#pragma unroll
    for (int i = 0; i < 9; i++) M[i] = Rf[i];
  } else {
    float Ki[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Ki[i] = e.Ki[i];
    mat3f_mul(Rf, Ki, M);
  }
  if (st) {
#pragma unroll
    for (int i = 0; i < 9; i++) e.M[i] = M[i];
    e.t[0] = (float)pose[4];
    e.t[1] = (float)pose[5];
    e.t[2] = (float)pose[6];
  }
}
// affLL = fromToVecExposure(lastRef->ab_exposure, new_frame_->ab_exposure, lastRef_aff_g2l, aff_g2l) (:717-720); the loop-closure
// estimator's reference affine is (0, 0) (PoseEstimator.cpp:317)
__device__ __forceinline__ void make_eval_aff(const TrackerDev &T, EvalIn &e, int mode, const double aff[2], bool st) {
  double affd[2];
  aff_from_to(T.ref_exposure, T.exposure[0], mode == 2 ? 0.0 : T.ref_a, mode == 2 ? 0.0 : T.ref_b, aff[0], aff[1], affd);
  if (st) {
    e.aff0 = (float)affd[0];
    e.aff1 = (float)affd[1];
  }
}
__device__ __forceinline__ void make_eval_cutoff(const TrackerDev &T, EvalIn &e, float cutoff, bool residual_only, bool st) {
  const float h = T.p.huber_th;
  if (st) {
    e.cutoff = cutoff;
    e.max_energy = 2 * h * cutoff - h * h; // :726-728 / :1030-1032
    e.residual_only = residual_only ? 1 : 0;
  }
}

// The complete inputs of an evaluation at `lvl` into the problem state: one thread (a level's first evaluation, a single evaluation
// of dsm_tracker_calc_res_*).
__device__ __forceinline__ void make_eval_any(const TrackerDev &T, LMState &S, int mode, int lvl, const double pose[7], const double aff[2],
                              float scale, float cutoff) {
  EvalIn &e = S.in;
  make_eval_level(T, e, mode, lvl);
  if (mode == 1) {
    e.scale = scale;
  } else {
    make_eval_rot(e, mode, pose, true);
    make_eval_aff(T, e, mode, aff, true);
  }
  make_eval_cutoff(T, e, cutoff, false, true);
}

// lane 0 only
__device__ __forceinline__ void begin_level(const TrackerDev &T, LMState &S, int lvl) {
  S.lvl = lvl;
  S.phase = PH_INIT;
  S.iteration = 0;
  S.level_cutoff_repeat = 1.0f;
  S.spec_valid = 0;
  const float cutoff = T.p.coarse_cutoff_th * S.level_cutoff_repeat;
  make_eval_any(T, S, S.is_scale, lvl, S.cur, S.aff_cur, S.scale_cur, cutoff);
  make_eval_level(T, S.spec_in, S.is_scale, lvl); // (the speculative candidate's block: a proposal writes its own part only)
}

// iteration bound of a level: maxIterations[lvl] (:463,:505 / :862,:897), or the benchmark schedule's K (dsm_params.fixed_schedule)
__device__ __forceinline__ int level_max_it(const ParamsDev &p, int lvl) { return p.fixed_schedule > 0 ? p.fixed_schedule : p.max_iterations[lvl]; }

// Vec6 rs of calcResPose / calcResScale (:843-851) from the reduced sums
// The E slot of the partials holds the energy of the USABLE lanes; every saturated lane contributes the evaluation's constant
// max_energy (:800 / :809), added here as n_sat x max_energy (no per-point select for it in the loop).
__device__ __forceinline__ void build_rs(const double *sums, const long long *isums, float max_energy, double rs[6]) {
  const float E = (float)(sums[kSlotE] + (double)max_energy * (double)isums[1]);
  const float sT = (float)sums[kSlotFlowT], sRT = (float)sums[kSlotFlowRT], sN = (float)sums[kSlotFlowNum];
  const int n_terms = (int)isums[0], n_sat = (int)isums[1];
  rs[0] = E;
  rs[1] = n_terms;
  rs[2] = sT / (sN + 0.1);
  rs[3] = 0;
  rs[4] = sRT / (sN + 0.1);
  rs[5] = n_sat / (float)n_terms;
}

__device__ __forceinline__ double scale_of(const ParamsDev &p, int i) { // SCALE_* per tangent index (:685-696)
  return i < 3 ? (double)p.scale_xi_rot : i < 6 ? (double)p.scale_xi_trans : i == 6 ? (double)p.scale_a : (double)p.scale_b;
}

// index of (r,c), r<=c, in the row-major upper triangle of the 9x9 accumulator
__device__ __forceinline__ int tri_idx(int r, int c) { return r * 9 - (r * (r - 1)) / 2 + (c - r); }

// H_out(r,c) of calcGSSSEPose (:681-692) for this lane's (r,c); b_out(r) (:683,:693-696)
__device__ __forceinline__ double build_H_elem(const ParamsDev &p, const double *sums, int n4, int r, int c) {
  const float hf = (float)sums[r <= c ? tri_idx(r, c) : tri_idx(c, r)];
  const float invn = 1.0f / n4; // (1.0f / n), n padded to a multiple of 4 (quirk Q3)
  return (((double)hf * (double)invn) * scale_of(p, c)) * scale_of(p, r);
}
__device__ __forceinline__ double build_b_elem(const ParamsDev &p, const double *sums, int n4, int r) {
  const float hf = (float)sums[tri_idx(r, 8)];
  const float invn = 1.0f / n4;
  return ((double)hf * (double)invn) * scale_of(p, r);
}

// Eigen LDLT<Lower> + solve (call sites :509,:513,:518,:529): the unblocked in-place algorithm restated from its published
// form -- the same restatement the CPU checker (orc_ldlt_solve) carries, and it is THAT checker the
// increments are bit-identical to.  Real Eigen is not in this image and was never run against either: its A21 update goes
// through the gemv kernel (column blocks of four, alpha = -1 folded into the accumulation), so last-bit differences from the
// reference's own solve are possible; the pivot-order argument below does not depend on that.  Executed by one wave on
// wave-uniform data.
//
// Eigen's unblocked LDLT is LEFT-looking: at step k it looks for the first maximum of |diagonal| among the rows not yet
// eliminated, swaps it into position k, and only then applies the pending updates to column k.  The diagonal entries it
// compares have therefore not been touched by the elimination: the whole pivot order is a function of the input diagonal
// alone (a selection sort with Eigen's swap bookkeeping).  That allows a form with a short dependent chain:
//   1. the pivot order from ONE all-pairs comparison of the diagonal (lane (r,c) tests |d_c| > |d_r|; a row's rank is the
//      population count of its byte of the ballot).  Exact ties and NaNs -- where Eigen's swaps, not the ranks, decide --
//      take a literal, sequential restatement of the selection loop instead (wave-uniform rare branch);
//   2. lane i reads row i of the symmetrically permuted lower triangle from the LDS copy of H;
//   3. the factorisation and the substitution passes run fully unrolled with static register indices, one ROW per lane:
//      the d_j A[k][j] products of a step are wave-uniform (lane reads), each lane forms the dot product over its own row
//      -- the update of A(k,k) in lane k, of column k's entries in the lanes below -- and the column is divided by the pivot
//      with ONE division per step; no pivot search and no divergence inside the chain; L^-T gets its columns through a
//      64-word LDS transpose;
//   4. the solution is un-permuted through eight LDS words.
// Every element sees the checker's operations in the checker's order: given bitwise equal H and b the increments are bitwise
// equal to the checker's (not: proven equal to Eigen's, see above).  (The fully wave-uniform form of the same arithmetic -- every lane all 36 entries --
// took 6.8-7.7 k shader cycles, issue-bound by its 36 IEEE double divisions; the cross-lane right-looking form of rounds
// 1-2, another pivot order, 7.7 k.)
// Rows / columns whose bit is clear in `active` do not take part (the 6- and 7-dim sub-solves): they are ordered last and
// enter as zero rows, zero columns and a zero right-hand side, which leaves every operation on the active block unchanged
// (x - 0 * y = x) and yields 0 for them.  stitch: row / column 6 of the system is row / column 7 of H (:521-534).
struct LdltScratch {
  double x[8];
  double av[8];
  double L[8][8]; // the factor, for its transpose
  int ix[8];
  int perm[8];
};

__device__ __forceinline__ void wave_ldlt_solve8(const double *Hlds, const double *blds, float lambda, unsigned active, bool stitch,
                                                 int lane, LdltScratch &scr, double inc[8]) {
  const int r = lane >> 3, c = lane & 7;
  const double lam1 = (double)(1 + lambda); // Hl(i,i) *= (1 + lambda): a float sum, widened (:506-508)
  auto src = [stitch](int i) { return stitch && i == 6 ? 7 : i; };
  const int nact = __builtin_popcount(active & 0xFFu);
  // ---- 1. pivot order ----
  int perm[8];
  {
    const double ar = fabs(Hlds[9 * src(r)] * lam1), ac = fabs(Hlds[9 * src(c)] * lam1);
    const bool act_r = (active >> r) & 1u, act_c = (active >> c) & 1u;
    const unsigned long long gt = __ballot(act_r && act_c && ac > ar);
    const bool odd = act_r && ((act_c && r != c && ac == ar) || ar != ar);
    const bool slow = __ballot(odd) != 0ull; // wave-uniform
    const int rank = act_r ? __builtin_popcount((unsigned)(gt >> (8 * r)) & 0xFFu) : nact + __builtin_popcount(~active & ((1u << r) - 1u) & 0xFFu);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned long long m = __ballot(c == 0 && rank == k);
      perm[k] = m ? (__builtin_ctzll(m) >> 3) : k;
    }
    if (__builtin_expect(slow, 0)) {
      // Eigen's selection loop as written: first maximum of the remaining diagonal IN ITS CURRENT ARRANGEMENT, swapped to
      // position k (comparisons with NaN are false: a NaN is only ever taken where it already stands)
      if (lane == 0) {
        int n = 0;
        for (int i = 0; i < 8; i++)
          if ((active >> i) & 1u) {
            scr.av[n] = fabs(Hlds[9 * src(i)] * lam1);
            scr.ix[n] = i;
            n++;
          }
        for (int k = 0; k < n; k++) {
          int big = k;
          double bigv = scr.av[k];
          for (int i = k + 1; i < n; i++)
            if (scr.av[i] > bigv) bigv = scr.av[i], big = i;
          const double tv = scr.av[k];
          const int ti = scr.ix[k];
          scr.av[k] = scr.av[big], scr.ix[k] = scr.ix[big];
          scr.av[big] = tv, scr.ix[big] = ti;
        }
        for (int i = 0; i < 8; i++)
          if (!((active >> i) & 1u)) scr.ix[n++] = i;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int k = 0; k < 8; k++) perm[k] = __builtin_amdgcn_readfirstlane(scr.ix[k]);
    }
  }
  // ---- 2. the permuted system, ONE ROW PER LANE (row i = lane & 7, replicated in every group of eight lanes): lower
  // triangle of P A P^T (LDLT<Lower> reads the lower triangle of its input), P rhs ----
  const int i = lane & 7;
  int pl = perm[0]; // this lane's unknown (logical index)
#pragma unroll
  for (int k = 1; k < 8; k++) pl = i == k ? perm[k] : pl;
  const int pi = src(pl);
  const bool act_i = i < nact;
  double a[8]; // a[j] = A[i][j], j <= i
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int pj = src(perm[j]);
    const int hi = pi > pj ? pi : pj, lo = pi > pj ? pj : pi;
    const double v = Hlds[8 * hi + lo];
    const double vd = v * lam1;
    a[j] = (act_i && j <= i) ? (i == j ? vd : v) : 0.0; // (j <= i: j active whenever i is)
  }
  double y = act_i ? -blds[pi] : 0.0;
  // ---- 3. factorisation, left-looking (no pivoting left to do), Eigen's operation order.  At step k every lane forms
  // s = sum_j A[i][j] temp[j] over ITS row -- the dot product of A(k,k)'s update in lane k, the A21 update in the lanes below
  // -- from the wave-uniform temp[j] = d_j A[k][j] (lane reads); ONE division per step instead of 7 - k. ----
  double du[8]; // the diagonal D, wave-uniform
  double d_own = 0.0;
  bool all_zero = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k > 0) {
      double sacc = 0;
#pragma unroll
      for (int j = 0; j < k; j++) {
        const double temp = du[j] * lane_value_d(a[j], k);
        sacc += a[j] * temp;
      }
      a[k] = a[k] - sacc;
    }
    const double akk = lane_value_d(a[k], k);
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) all_zero = true;
    const double q = a[k] / akk;
    a[k] = (i > k && valid) ? q : a[k];
    du[k] = akk;
    d_own = i == k ? akk : d_own;
  }
  // L^-1: y[i] -= A[i][j] y[j], j ascending (lane j's value is final when its turn comes)
#pragma unroll
  for (int j = 0; j < 7; j++) {
    const double yj = lane_value_d(y, j);
    y = i > j ? y - a[j] * yj : y;
  }
  {
    const double tol = 1.0 / 1.7976931348623157e308; // Eigen: 1 / NumTraits<double>::highest()
    y = fabs(d_own) > tol ? y / d_own : 0.0;
  }
  // L^-T needs column i of L in lane i: transpose through LDS
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) scr.L[i][j] = a[j];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double cl[8];
#pragma unroll
  for (int j = 0; j < 8; j++) cl[j] = scr.L[j][i]; // A[j][i]
  // y[i] -= A[j][i] y[j] for i = 6 .. 0, j = i + 1 .. 7 ascending (Eigen's order); x_j is final once row j is done
  double xu[8];
  xu[7] = lane_value_d(y, 7);
#pragma unroll
  for (int ii = 6; ii >= 0; ii--) {
#pragma unroll
    for (int j = ii + 1; j < 8; j++) y = i == ii ? y - cl[j] * xu[j] : y;
    xu[ii] = lane_value_d(y, ii);
  }
  // ---- 4. P^T: row i of the permuted system is unknown pl ----
  if (lane < 8) scr.x[pl] = (all_zero || !act_i) ? 0.0 : y;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
  for (int k = 0; k < 8; k++) inc[k] = scr.x[k];
  if (stitch) {
    inc[7] = inc[6];
    inc[6] = 0;
  }
}

// value of an LDS word that another wave of the workgroup writes
__device__ __forceinline__ int lds_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_set(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A proposing wave's HELPER (round 6).  Of what follows the solve, the affine part of the next evaluation's inputs -- affLL: the
// step's only exp, a division -- and its cut-off need nothing of the new pose: a wave that would otherwise sit idle forms them while
// the proposing wave runs SE3::exp, the pose product and R K^-1.  The proposing wave posts the candidate's affine pair as soon as the
// increment is known, and meets the helper again before it leaves.
struct LmHelp {
  int cmd;  // proposing wave -> helper: 0 wait, 1 go, 2 nothing to do this step
  int done; // helper -> proposing wave
  int spec, last;
  float cutoff, pad;
  double aff[2];
};
constexpr int kLmSpinBound = 1 << 22; // polls of an LDS flag before a wave gives up (seconds; a step is microseconds): waits are bounded
__device__ int g_lm_spin_expired;      // ... and say so: nonzero once any bounded wait of an LM step has expired (read by the host with the statistics)
__device__ __forceinline__ int lds_wait_nonzero(const int *p) {
  int v = 0;
  for (int spins = 0; (v = lds_flag(p)) == 0; spins++) {
    if (spins > kLmSpinBound) {
      atomicAdd(&g_lm_spin_expired, 1);
      return -1;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return v;
}
__device__ __forceinline__ void lm_help_wave(const TrackerDev &T, LMState &S, LmHelp &h, int lane) {
  const int cmd = lds_wait_nonzero(&h.cmd);
  if (cmd != 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  EvalIn &e = h.spec ? S.spec_in : S.in;
  const double aff[2] = {h.aff[0], h.aff[1]};
  make_eval_aff(T, e, S.is_scale, aff, lane == 0);
  make_eval_cutoff(T, e, h.cutoff, h.last != 0, lane == 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) lds_flag_set(&h.done, 1);
}

// lane 0 only: :612-637
__device__ __forceinline__ void finish_track(const TrackerDev &T, LMState &S) {
  const float modeA = T.p.affine_opt_mode_a, modeB = T.p.affine_opt_mode_b;
  int status = ST_GOOD;
  if ((modeA != 0 && (__builtin_fabsf((float)S.aff_cur[0]) > 1.2)) ||
      (modeB != 0 && (__builtin_fabsf((float)S.aff_cur[1]) > 200)))
    status = ST_BAD_AFFINE;
  if (status == ST_GOOD) {
    double rel[2];
    aff_from_to(T.ref_exposure, T.exposure[0], T.ref_a, T.ref_b, S.aff_cur[0], S.aff_cur[1], rel);
    const float rel0 = (float)rel[0], rel1 = (float)rel[1];
    if ((modeA == 0 && (__builtin_fabsf(logf(rel0)) > 1.5)) || (modeB == 0 && (__builtin_fabsf(rel1) > 200)))
      status = ST_BAD_AFFINE;
  }
  if (status == ST_GOOD) {
    if (modeA < 0) S.aff_cur[0] = 0;
    if (modeB < 0) S.aff_cur[1] = 0;
  }
  S.status = status;
}

// whole wave: solve + propose for the pose problem (:505-554) from the H, b of the problem state (LDS).
// spec: the proposal is the speculative one (S.spec_*).  hp: this wave's helper (lm_help_wave), or null.
__device__ __forceinline__ void propose_pose(const TrackerDev &T, LMState &S, float lambda, int lane, LdltScratch &scr, bool spec = false,
                                             LmHelp *hp = nullptr) {
  const float modeA = T.p.affine_opt_mode_a, modeB = T.p.affine_opt_mode_b;
  unsigned active = 0xFFu;
  bool stitch = false;
  if (modeA < 0 && modeB < 0) { // :511-515 fix a, b
    active = 0x3Fu;
  } else if (!(modeA < 0) && modeB < 0) { // :516-520 fix b
    active = 0x7Fu;
  } else if (modeA < 0 && !(modeB < 0)) { // :521-534 fix a: row/col 6 := row/col 7
    stitch = true;
    active = 0x7Fu;
  }
  double inc[8];
  wave_ldlt_solve8(S.H, S.b, lambda, active, stitch, lane, scr, inc);
  // From here on every lane computes the same (wave-uniform) values; lane 0 stores them.  The two
  // sincos evaluations of SE3::exp run side by side in lanes 0 and 1.
  float extrapFac = 1; // :536-539
  const float lim = T.p.lambda_extrapolation_limit;
  if (lambda < lim) extrapFac = sqrtf(sqrtf(lim / lambda));
#pragma unroll
  for (int i = 0; i < 8; i++) inc[i] *= extrapFac;
  double incScaled[8], sum = 0; // :541-548
#pragma unroll
  for (int i = 0; i < 8; i++) {
    incScaled[i] = inc[i] * scale_of(T.p, i);
    sum += incScaled[i];
  }
  if (!__builtin_isfinite(sum)) {
#pragma unroll
    for (int i = 0; i < 8; i++) incScaled[i] = 0;
  }
  double aff_cand[2] = {S.aff_cur[0] + incScaled[6], S.aff_cur[1] + incScaled[7]};
  double nrm = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i];
  const double inc_norm = sqrt(nrm);
  // Is this the loop's last evaluation?  The step that consumes it ends the level when the increment is small (:588) or the
  // iteration bound is reached (:505) -- both known now; the speculative proposal is consumed one iteration later.
  const bool last = (T.p.fixed_schedule <= 0 && !(inc_norm > 1e-3)) || S.iteration + (spec ? 2 : 1) >= level_max_it(T.p, S.lvl);
  const float cutoff = T.p.coarse_cutoff_th * S.level_cutoff_repeat;
  EvalIn &e = spec ? S.spec_in : S.in; // (its level part is in place since begin_level)
  if (hp) { // the affine pair and the cut-off go to the helper now; the pose part follows here
    if (lane == 0) {
      hp->aff[0] = aff_cand[0], hp->aff[1] = aff_cand[1];
      hp->cutoff = cutoff;
      hp->last = last ? 1 : 0;
      hp->spec = spec ? 1 : 0;
      lds_flag_set(&hp->cmd, 1);
    }
  }
  double ex[7], cur[7], cand[7];
#pragma unroll
  for (int i = 0; i < 7; i++) cur[i] = S.cur[i];
  se3_exp_wave(incScaled, ex, lane);
  se3_mul(ex, cur, cand); // :550-551
  if (lane == 0) {
    if (spec) {
#pragma unroll
      for (int i = 0; i < 7; i++) S.spec_cand[i] = cand[i];
      S.spec_aff_cand[0] = aff_cand[0];
      S.spec_aff_cand[1] = aff_cand[1];
      S.spec_inc_norm = inc_norm;
    } else {
#pragma unroll
      for (int i = 0; i < 7; i++) S.cand[i] = cand[i];
      S.aff_cand[0] = aff_cand[0];
      S.aff_cand[1] = aff_cand[1];
      S.inc_norm = inc_norm;
      S.phase = PH_ITER;
    }
  }
  make_eval_rot(e, S.is_scale, cand, lane == 0);
  if (hp) {
    (void)lds_wait_nonzero(&hp->done);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    make_eval_aff(T, e, S.is_scale, aff_cand, lane == 0);
    make_eval_cutoff(T, e, cutoff, last, lane == 0);
  }
}

// lane 0 only: :897-913
__device__ __forceinline__ void propose_scale(const TrackerDev &T, LMState &S, float lambda, bool spec = false) {
  float Hl = S.Hs;
  Hl *= (1 + lambda);
  float inc = -S.bs / Hl;
  float extrapFac = 1;
  const float lim = T.p.lambda_extrapolation_limit;
  if (lambda < lim) extrapFac = sqrtf(sqrtf(lim / lambda));
  inc *= extrapFac;
  if (!__builtin_isfinite(inc) || __builtin_fabsf(inc) > S.scale_cur) inc = 0.0f;
  const float cand = S.scale_cur + inc;
  if (spec) {
    S.spec_inc_f = inc;
    S.spec_scale_cand = cand;
  } else {
    S.inc_f = inc;
    S.scale_cand = cand;
    S.phase = PH_ITER;
  }
  const bool last = (T.p.fixed_schedule <= 0 && !(inc > 1e-3)) || S.iteration + (spec ? 2 : 1) >= level_max_it(T.p, S.lvl); // :937 (signed, Q7) / :897
  EvalIn &e = spec ? S.spec_in : S.in; // (R10 K^-1, tsl_f1_f0 and the camera are the level's: in place since begin_level)
  e.scale = cand;
  make_eval_cutoff(T, e, T.p.coarse_cutoff_th * S.level_cutoff_repeat, last, true);
}

// lane 0 only
__device__ __forceinline__ void end_level(const TrackerDev &T, LMState &S) {
  const int lvl = S.lvl;
  S.last_residuals[lvl] = sqrtf((float)(S.res_old[0] / S.res_old[1])); // :596 / :945
  S.last_inners[lvl] = S.res_old[1];                                     // PoseEstimator.cpp:463
  if (S.is_scale == 0) {
    S.flow[0] = S.res_old[2]; // :597
    S.flow[1] = S.res_old[3];
    S.flow[2] = S.res_old[4];
    if (T.p.fixed_schedule <= 0 && S.last_residuals[lvl] > 1.5 * S.min_res[lvl]) { // :598
      S.status = ST_ABORTED;
      return;
    }
  }
  int next = lvl - 1;
  if (S.level_cutoff_repeat > 1 && !S.have_repeated) { // :601-604 / :947-950
    next = lvl;
    S.have_repeated = 1;
  }
  if (next < 0) {
    if (S.is_scale == 1)
      S.status = ST_GOOD;
    else
      finish_track(T, S); // the same affine plausibility checks end PoseEstimator::estimate (:470-482)
    return;
  }
  begin_level(T, S, next);
}

// LDS workspace of the partial reduction / LM step (lm_kernel and coarse_kernel)
struct RedBuf { // fixed-order reduction of one evaluation's chunk partials
  double psum[19][kNumSlots];
  long long pisum[19][4];
  double sums[kNumSlots];
  long long isums[4];
};
// second reduction buffer and the wave 0 <-> wave 1 hand-shake of a step that handles a speculative candidate
struct LmSpecShared {
  RedBuf red;
  LdltScratch ldlt; // wave 1's
  LmHelp help;      // wave 1's helper (wave 3)
  int cmd;      // wave 0 -> wave 1: 0 wait, 1 stage the speculative proposal with `lambda`, 2 nothing to do
  int done;     // wave 1 -> wave 0
  float lambda;
};
struct LmShared {
  RedBuf red;
  LdltScratch ldlt; // wave 0's
  LmHelp help;      // wave 0's helper (wave 2)
  // The problem's LMState and its tracker descriptor are staged here for the duration of a step
  // (lm_kernel) or of the whole small-level loop (coarse_kernel): the state machine then runs on LDS
  // latencies instead of a chain of dependent global-memory round trips.
  __attribute__((aligned(16))) LMState st;
  __attribute__((aligned(16))) TrackerDev trk;
};
static_assert(sizeof(LMState) % 16 == 0 && sizeof(TrackerDev) % 16 == 0, "staged with 16-byte copies");

// 16-byte block copies between global memory and LDS by `nthreads` threads
template <class T>
__device__ __forceinline__ void stage_in(T &dst_lds, const T *src_global, int tid, int nthreads) {
  constexpr int n16 = sizeof(T) / 16;
  const uint4 *src = (const uint4 *)src_global;
  uint4 *dst = (uint4 *)&dst_lds;
  for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
}
template <class T>
__device__ __forceinline__ void stage_out(T *dst_global, const T &src_lds, int tid, int nthreads) {
  constexpr int n16 = sizeof(T) / 16;
  const uint4 *src = (const uint4 *)&src_lds;
  uint4 *dst = (uint4 *)dst_global;
  for (int i = tid; i < n16; i += nthreads) dst[i] = src[i];
}

// Device-coherent 16-byte accesses (two 8-byte device-scope atomics): for state that one workgroup writes and a
// workgroup on another XCD reads inside the SAME launch (work-queue kernel); the XCD L2s are not coherent with
// each other for ordinary accesses.
__device__ __forceinline__ uint4 load16_coherent(const void *p) {
  const unsigned long long a = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load((const unsigned long long *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint4 v;
  v.x = (unsigned)a, v.y = (unsigned)(a >> 32), v.z = (unsigned)b, v.w = (unsigned)(b >> 32);
  return v;
}
__device__ __forceinline__ void store16_coherent(void *p, const uint4 &v) {
  __hip_atomic_store((unsigned long long *)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long *)p + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- fixed-order reduction over the chunk partials (double / int64), first 247 threads of the
// workgroup.  Thread (g, q) sums the slot quad q (one float4 = 4 of the 52 slots) over chunks
// g, g+19, g+38, ...: every load is a 16-byte read and up to kRedBatch of them are in flight per
// thread.  `P` may point to global memory (lm_kernel) or LDS (coarse_kernel): same order, same sums.
// reduce_partials_groups2: TWO evaluations' partials -- the main and the speculative candidate's -- reduced side by side, each in
// exactly the order above, their loads requested TOGETHER (two reductions one after the other were two memory round trips in front of
// every LM step of the small levels; shader-clock stamps: profiles/r06_tick_stamps.json).  A speculative candidate exists on levels of at
// most 8192 points only, i.e. at most 32 chunks = two per group: a batch of two is the whole job, and the registers stay few -- an LM
// kernel must fit the hole ONE retiring evaluation wave leaves on a SIMD (128 VGPRs): with 138 the scale segment's LM launches, which
// run beside the pose evaluations, took 68 us instead of 12 (profiles/r06_lm_vgpr_regression.txt).
__device__ __forceinline__ void reduce_partials_groups(const float *P, int nch, int tid, RedBuf &sh) {
  constexpr int kGroups = 19, kQuads = kNumSlots / 4, kRedBatch = 8;
  const int q = tid % kQuads, g = tid / kQuads;
  if (g < kGroups) {
    double sd[4] = {0, 0, 0, 0};
    long long si[4] = {0, 0, 0, 0};
    for (int c0 = g; c0 < nch; c0 += kGroups * kRedBatch) {
      fvec4 v[kRedBatch];
#pragma unroll
      for (int j = 0; j < kRedBatch; j++) {
        const int cc = c0 + j * kGroups;
        v[j] = cc < nch ? load_partial4(P + ((size_t)cc * (kPartialStride / 4) + q) * 4) : fvec4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < kRedBatch; j++) {
        sd[0] += (double)v[j].x, sd[1] += (double)v[j].y, sd[2] += (double)v[j].z, sd[3] += (double)v[j].w;
        si[0] += __float_as_int(v[j].x), si[1] += __float_as_int(v[j].y), si[2] += __float_as_int(v[j].z),
            si[3] += __float_as_int(v[j].w);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int slot = 4 * q + e;
      if (slot < kSlotNTerms)
        sh.psum[g][slot] = sd[e];
      else
        sh.pisum[g][slot - kSlotNTerms] = si[e];
    }
  }
}
__device__ __forceinline__ void reduce_partials_groups2(const float *P, const float *P2, int nch, int tid, RedBuf &sh, RedBuf &sh2) {
  constexpr int kGroups = 19, kQuads = kNumSlots / 4, kRedBatch = 2;
  const int q = tid % kQuads, g = tid / kQuads;
  if (g < kGroups) {
    double sd[4] = {0, 0, 0, 0}, td[4] = {0, 0, 0, 0};
    long long si[4] = {0, 0, 0, 0}, ti[4] = {0, 0, 0, 0};
    for (int c0 = g; c0 < nch; c0 += kGroups * kRedBatch) {
      fvec4 v[kRedBatch], w[kRedBatch];
#pragma unroll
      for (int j = 0; j < kRedBatch; j++) {
        const int cc = c0 + j * kGroups;
        v[j] = cc < nch ? load_partial4(P + ((size_t)cc * (kPartialStride / 4) + q) * 4) : fvec4{0.f, 0.f, 0.f, 0.f};
        w[j] = cc < nch ? load_partial4(P2 + ((size_t)cc * (kPartialStride / 4) + q) * 4) : fvec4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < kRedBatch; j++) {
        sd[0] += (double)v[j].x, sd[1] += (double)v[j].y, sd[2] += (double)v[j].z, sd[3] += (double)v[j].w;
        si[0] += __float_as_int(v[j].x), si[1] += __float_as_int(v[j].y), si[2] += __float_as_int(v[j].z),
            si[3] += __float_as_int(v[j].w);
        td[0] += (double)w[j].x, td[1] += (double)w[j].y, td[2] += (double)w[j].z, td[3] += (double)w[j].w;
        ti[0] += __float_as_int(w[j].x), ti[1] += __float_as_int(w[j].y), ti[2] += __float_as_int(w[j].z),
            ti[3] += __float_as_int(w[j].w);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int slot = 4 * q + e;
      if (slot < kSlotNTerms)
        sh.psum[g][slot] = sd[e], sh2.psum[g][slot] = td[e];
      else
        sh.pisum[g][slot - kSlotNTerms] = si[e], sh2.pisum[g][slot - kSlotNTerms] = ti[e];
    }
  }
}

// wave 0, after a workgroup barrier: the 19 group sums are added in group order by the slot's lane
__device__ __forceinline__ void reduce_partials_final(int lane, RedBuf &sh) {
  if (lane < kSlotNTerms) {
    double s = sh.psum[0][lane];
#pragma unroll
    for (int g = 1; g < 19; g++) s += sh.psum[g][lane];
    sh.sums[lane] = s;
  } else if (lane < kNumSlots) {
    long long s = 0;
#pragma unroll
    for (int g = 0; g < 19; g++) s += sh.pisum[g][lane - kSlotNTerms];
    sh.isums[lane - kSlotNTerms] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // LDS writes above are read by other lanes below
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}


// One step of the LM state machine by wave 0 (all 64 lanes; lane 0 writes the state).
// sp != nullptr: the launch also evaluated the speculative candidates (where S.spec_valid) -- their reduced sums are in
// sp->red -- and new proposals may stage one; wave 1 of the workgroup runs lm_spec_wave1 beside this function.
// hp: wave 0's helper is running beside it (lm_help_wave on sh.help): it is told what to do, or that there is nothing, exactly once.
// early: called (whole wave) as soon as the step knows that it proposes -- i.e. that the next evaluation is one of THIS level -- and
// whether a speculative twin may be staged: before the solve and everything after it (the tick engine asks for its place in the item
// list then, a returning atomic whose round trip used to follow the step).
struct LmNoEarly {
  __device__ __forceinline__ void operator()(const LMState &, int, bool) const {}
};
template <class EARLY = LmNoEarly>
__device__ __forceinline__ void lm_step_wave0(int mode, int lvl, const TrackerDev &T, LMState &S, LmShared &sh, int lane,
                                              LmSpecShared *sp = nullptr, bool helped = false, EARLY *early = nullptr) {
  const bool pose_like = mode != 1;
  const double *sums = sh.red.sums;
  const long long *isums = sh.red.isums;
  double rs[6];
  build_rs(sums, isums, S.in.max_energy, rs);
  const int n_warped = (int)isums[2];
  const int n4 = (n_warped + 3) & ~3; // :824-835 padding counts in n (quirk Q3)
  const int r = lane >> 3, c = lane & 7;
  // ---- LM_OP_STEP: every lane of wave 0 evaluates the same (uniform) decisions; lane 0 writes ----
  const int max_it = level_max_it(T.p, lvl);
  const bool fixed = T.p.fixed_schedule > 0; // benchmark schedule: every step is taken, nothing ends a level but the bound
  const float lim = T.p.lambda_extrapolation_limit;
  const int phase = S.phase;
  const bool had_spec = sp && S.spec_valid; // wave-uniform; read before lane 0 clears it
  bool level_done = false, do_propose = false;
  float lambda_next = 0.01f; // computed by every lane: nothing lane 0 writes below is re-read by the wave
  int iteration = 0;
  if (lane == 0) {
    S.rounds[lvl]++;
    S.ticks++; // (a chain of the tick engine takes its extra rounds off again)
    S.spec_valid = 0;
  }
  if (phase == PH_INIT) {
    if (!fixed && rs[5] > 0.6 && S.level_cutoff_repeat < 50) { // :477-485 / :875-883
      if (lane == 0) {
        S.evals[lvl]++;
        S.level_cutoff_repeat *= 2;
        const float cutoff = T.p.coarse_cutoff_th * S.level_cutoff_repeat;
        make_eval_cutoff(T, S.in, cutoff, false, true); // (the same pose or scale once more, with the wider cut-off: nothing else changes)
      }
    } else {
      if (pose_like) {
        S.H[lane] = build_H_elem(T.p, sums, n4, r, c); // :487
        if (c == 0) S.b[r] = build_b_elem(T.p, sums, n4, r);
      }
      if (lane == 0) {
        S.evals[lvl]++;
        for (int i = 0; i < 6; i++) S.res_old[i] = rs[i];
        if (!pose_like) {
          S.Hs = (float)sums[0] * (1.0f / n4); // :885
          S.bs = (float)sums[1] * (1.0f / n4);
        }
        S.lambda = lambda_next; // 0.01, :489
        S.iteration = 0;
      }
      if (max_it <= 0)
        level_done = true;
      else
        do_propose = true;
    }
  } else {
    const double old_ratio = S.res_old[0] / S.res_old[1];
    const bool accept = fixed || (rs[0] / rs[1]) < old_ratio; // :559 / :915
    const bool small = !fixed && (pose_like ? !(S.inc_norm > 1e-3) : !(S.inc_f > 1e-3)); // :588 / :937 (signed, Q7)
    iteration = S.iteration + 1;
    {
      const float l_old = S.lambda;
      float l4 = l_old * 4;
      if (l4 < lim) l4 = lim;
      lambda_next = accept ? l_old * 0.5f : l4; // :581 / :583-585
    }
    if (pose_like && accept) { // :576-577
      S.H[lane] = build_H_elem(T.p, sums, n4, r, c);
      if (c == 0) S.b[r] = build_b_elem(T.p, sums, n4, r);
    }
    if (lane == 0) {
      S.evals[lvl]++;
      S.evals_ro[lvl] += S.in.residual_only;
      if (accept) { // :576-581 / :926-930
        for (int i = 0; i < 6; i++) S.res_old[i] = rs[i];
        if (pose_like) {
          for (int i = 0; i < 7; i++) S.cur[i] = S.cand[i];
          S.aff_cur[0] = S.aff_cand[0];
          S.aff_cur[1] = S.aff_cand[1];
        } else {
          S.Hs = (float)sums[0] * (1.0f / n4);
          S.bs = (float)sums[1] * (1.0f / n4);
          S.scale_cur = S.scale_cand;
        }
      }
    }
    if (small || iteration >= max_it) {
      level_done = true;
    } else if (!accept && had_spec) {
      // The proposal the loop makes next -- same H, b and current pose, lambda = lambda_next -- is the speculative candidate:
      // it was evaluated in this launch.  Consume it as the loop's next iteration (:505-590 once more).
      const double *sums2 = sp->red.sums;
      const long long *isums2 = sp->red.isums;
      double rs2[6];
      build_rs(sums2, isums2, S.spec_in.max_energy, rs2);
      const int n4b = ((int)isums2[2] + 3) & ~3;
      const bool accept2 = (rs2[0] / rs2[1]) < old_ratio; // res_old is unchanged by the rejection
      const bool small2 = pose_like ? !(S.spec_inc_norm > 1e-3) : !(S.spec_inc_f > 1e-3);
      iteration += 1;
      {
        const float l_old = lambda_next;
        float l4 = l_old * 4;
        if (l4 < lim) l4 = lim;
        lambda_next = accept2 ? l_old * 0.5f : l4;
      }
      if (pose_like && accept2) {
        S.H[lane] = build_H_elem(T.p, sums2, n4b, r, c);
        if (c == 0) S.b[r] = build_b_elem(T.p, sums2, n4b, r);
      }
      if (lane == 0) {
        S.evals[lvl]++;
        S.evals_ro[lvl] += S.spec_in.residual_only;
        if (accept2) {
          for (int i = 0; i < 6; i++) S.res_old[i] = rs2[i];
          if (pose_like) {
            for (int i = 0; i < 7; i++) S.cur[i] = S.spec_cand[i];
            S.aff_cur[0] = S.spec_aff_cand[0];
            S.aff_cur[1] = S.spec_aff_cand[1];
          } else {
            S.Hs = (float)sums2[0] * (1.0f / n4b);
            S.bs = (float)sums2[1] * (1.0f / n4b);
            S.scale_cur = S.spec_scale_cand;
          }
        }
      }
      if (small2 || iteration >= max_it)
        level_done = true;
      else
        do_propose = true;
    } else {
      do_propose = true;
    }
    if (lane == 0) {
      S.lambda = lambda_next;
      S.iteration = iteration;
    }
  }
  // make lane 0's state writes (lambda, cur, ...) visible to the whole wave (and to wave 1) before proposing
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // the speculative proposal: what the loop proposes after rejecting the main one -- lambda four times larger (:583-585);
  // it exists only if the loop would go on after that rejection (one more iteration allowed; the main step not "too small")
  const bool want_spec = sp && do_propose && !fixed && iteration + 1 < max_it;
  if (early && do_propose) (*early)(S, lane, want_spec);
  if (sp && lane == 0) {
    float l4 = lambda_next * 4;
    if (l4 < lim) l4 = lim;
    sp->lambda = l4;
    lds_flag_set(&sp->cmd, want_spec ? 1 : 2);
  }
  if (do_propose) {
    if (pose_like)
      propose_pose(T, S, lambda_next, lane, sh.ldlt, false, helped ? &sh.help : nullptr);
    else if (lane == 0)
      propose_scale(T, S, lambda_next);
  }
  if (helped && !(do_propose && pose_like) && lane == 0) lds_flag_set(&sh.help.cmd, 2);
  if (want_spec) {
    (void)lds_wait_nonzero(&sp->done);
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const bool main_small = pose_like ? !(S.inc_norm > 1e-3) : !(S.inc_f > 1e-3); // the loop breaks after the main step
      S.spec_valid = main_small ? 0 : 1;
    }
  }
  if (lane == 0 && level_done) end_level(T, S);
}

// wave 1 beside lm_step_wave0: stages the speculative proposal when wave 0 asks for it (helped: its own helper runs on sp.help)
__device__ __forceinline__ void lm_spec_wave1(int mode, const TrackerDev &T, LMState &S, LmSpecShared &sp, int lane, bool helped = false) {
  const int cmd = lds_wait_nonzero(&sp.cmd);
  if (cmd != 1 || mode == 1) {
    if (helped && lane == 0) lds_flag_set(&sp.help.cmd, 2);
    if (cmd != 1) return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const float lambda = sp.lambda;
  if (mode != 1) {
    propose_pose(T, S, lambda, lane, sp.ldlt, true, helped ? &sp.help : nullptr);
  } else if (lane == 0) {
    propose_scale(T, S, lambda, true);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) lds_flag_set(&sp.done, 1);
}

// One LM step by a workgroup of >= 256 threads on the problem's partials: state and tracker descriptor
// staged through LDS, fixed-order reduction, wave 0 advances the state machine and writes the state
// back.  Shared by lm_kernel (LM_OP_STEP) and by the last-arriving workgroup of a fused eval kernel.
// All threads of the workgroup must call it; S must be at this level (status RUNNING, lvl, mode).
// sp != nullptr: speculative candidates (their partials sit spec_off floats behind the main ones).
// What a caller already holds when it calls lm_step_block (round 6: the step's memory phase was three to four DEPENDENT round trips
// -- tracker pointer -> descriptor, point count of the pending evaluation -> partials, main then speculative partials -- 11 k of the
// 37 k cycles an LM workgroup of the tick engine lives, shader-clock stamps in profiles/r06_tick_stamps.json).  With the point
// count known up front the descriptor, the state block and every partial are requested together: one round trip.
struct LmPre {
  int n_lvl, ppt_lvl; // S.in.n / S.in.ppt of the pending evaluation
  bool have_sv;       // sv holds this thread's 16-byte block of the state (already loaded with the caller's own first reads)
  uint4 sv;
};
constexpr int kLmS16 = sizeof(LMState) / 16, kLmT16 = sizeof(TrackerDev) / 16;
template <bool COH = false, class EARLY = LmNoEarly>
__device__ __forceinline__ void lm_step_block(int mode, int lvl, int prob, const TrackerDev *Tg, LMState &S,
                                              const float *partials_prob, LmShared &sh, int tid, int *status_out,
                                              LmSpecShared *sp = nullptr, int spec_off = 0, const LmPre *pre = nullptr, EARLY *early = nullptr,
                                              bool help = true) {
  constexpr int kS16 = kLmS16, kT16 = kLmT16;
  static_assert(kS16 <= kThreads && kT16 <= kThreads, "one 16-byte block per thread");
  // one round trip: state block, tracker descriptor and the chunk partials together
  uint4 sv = {0, 0, 0, 0}, tv = {0, 0, 0, 0};
  if (pre && pre->have_sv)
    sv = pre->sv;
  else if (tid < kS16)
    sv = COH ? load16_coherent((const uint4 *)&S + tid) : ((const uint4 *)&S)[tid];
  if (tid < kT16) tv = ((const uint4 *)Tg)[tid];
  // the pending evaluation was built for this level
  const int n_lvl = pre ? pre->n_lvl : COH ? __hip_atomic_load(&S.in.n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S.in.n;
  const int ppt_lvl = pre ? pre->ppt_lvl : COH ? __hip_atomic_load(&S.in.ppt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S.in.ppt;
  if (sp) { // (garbage where no speculative candidate was evaluated: never looked at then)
    reduce_partials_groups2(partials_prob, partials_prob + spec_off, chunks_of(n_lvl, ppt_lvl), tid, sh.red, sp->red);
    if (tid == 0) sp->cmd = 0, sp->done = 0, sp->help.cmd = 0, sp->help.done = 0;
  } else {
    reduce_partials_groups(partials_prob, chunks_of(n_lvl, ppt_lvl), tid, sh.red);
  }
  if (tid == 0) sh.help.cmd = 0, sh.help.done = 0;
  if (tid < kS16) ((uint4 *)&sh.st)[tid] = sv;
  if (tid < kT16) ((uint4 *)&sh.trk)[tid] = tv;
  __syncthreads();
  // waves 2 and 3: the helpers of the proposing waves 0 and 1 (the pose problems' affine part, lm_help_wave)
  const bool helped = mode != 1 && help;
  if (sp && tid >= 64 && tid < 128) lm_spec_wave1(mode, sh.trk, sh.st, *sp, tid - 64, helped);
  if (helped && tid >= 128 && tid < 192) lm_help_wave(sh.trk, sh.st, sh.help, tid - 128);
  if (helped && sp && tid >= 192 && tid < 256) lm_help_wave(sh.trk, sh.st, sp->help, tid - 192);
  if (tid >= 64) return; // wave 0 carries on
  const int lane = tid;
  reduce_partials_final(lane, sh.red);
  if (sp) reduce_partials_final(lane, sp->red);
  lm_step_wave0(mode, lvl, sh.trk, sh.st, sh, lane, sp, helped, early);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // lane 0's LDS writes -> the whole wave
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (COH) {
    for (int i = lane; i < kS16; i += 64) store16_coherent((uint4 *)&S + i, ((const uint4 *)&sh.st)[i]);
  } else {
    stage_out(&S, sh.st, lane, 64);
  }
  if (lane == 0 && status_out) {
    status_out[2 * prob] = sh.st.status;
    status_out[2 * prob + 1] = sh.st.lvl;
  }
}

// ------------------------------------------------------------------------------------------
// eval kernel: grid (chunk slots, problems).
// LVL0 = true is the level-0 instantiation (adds the flow indicators); it is also the dominant kernel
// of the path and shows up under its own symbol in rocprofv3 kernel traces.
// FUSED = true: the workgroup of a problem that finishes last (ticket counter over ALL gridDim.x
// workgroups of the problem's row, so that every one of them has read the state before it is
// rewritten) also runs the LM step -- no separate lm_kernel launch for this evaluation.  The step
// reads the same partials in the same order: results are bit-identical to the two-kernel form.
// ------------------------------------------------------------------------------------------
// The plain form (levels >= 1 of the launch-per-step schedule) is held to 96 VGPRs = five waves per SIMD, which is worth
// more there than the two or three registers the allocator would otherwise take (no spills); the level-0 and fused forms
// need 104-109 and run four.
// ROSEL: 0 = full and residual-only evaluations alike (chosen per problem at run time); 1 = this launch takes the full
// evaluations only, 2 = the residual-only ones only -- an instantiation of its own, without the 45 accumulators: 8 waves
// per SIMD instead of 4.
template <int MODE, bool LVL0, bool FUSED, int ROSEL = 0>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(ROSEL == 2 ? 7 : (LVL0 || FUSED || MODE == 1) ? 4 : 5))) void eval_kernel(const TrackerDev *const *__restrict__ trackers,
                                                        const LMState *__restrict__ states,
                                                        float *__restrict__ partials,
                                                        int partial_stride, int lvl, int *__restrict__ tickets,
                                                        int *__restrict__ status_out, int spec_nprob, const int *__restrict__ rowmap) {
  // spec_nprob > 0: grid rows [spec_nprob, 2 spec_nprob) evaluate the problems' speculative candidates (where staged)
  // rowmap != nullptr: a COMPACT launch -- grid row r works on problem rowmap[r] (the host lists the problems still at this
  // level after a read-back: a launch over all problems of a batch costs ~1.6 ns per workgroup just to start and end the
  // idle ones, 100 us for a level-0 grid of 512 problems; a level's last rounds are needed by a few stragglers only)
  const bool cand = spec_nprob > 0 && (int)blockIdx.y >= spec_nprob;
  const int row = cand ? blockIdx.y - spec_nprob : blockIdx.y;
  const int prob = rowmap ? ((const DSM_GLOBAL int *)rowmap)[row] : row;
  // Uniform, read-only state: global address space so that it becomes scalar (s_load) reads -- and ALL of them are issued
  // before the first one is looked at: the head of the state and the evaluation inputs of this row's candidate arrive in
  // ONE memory round trip instead of a chain of three (state -> candidate -> inputs), which was a seventh of a mid-level
  // workgroup's life.  (The asm statements pin the loads above the early exits; they emit no instruction.)
  const DSM_GLOBAL LMState &S = ((const DSM_GLOBAL LMState *)states)[prob];
  const DSM_GLOBAL EvalIn &in = cand ? S.spec_in : S.in;
  const int s_status = S.status, s_lvl = S.lvl, s_kind = S.is_scale, s_spec_valid = S.spec_valid;
  const TrackerDev *trk_ptr = FUSED ? ((const TrackerDev *const DSM_GLOBAL *)trackers)[prob] : nullptr; // (the fused step's descriptor: asked for with the state)
  // the MAIN candidate's point count: what the fused step reduces over (a speculative row's own inputs, S.spec_in, are stale where no
  // candidate was staged -- another level's count)
  const int main_n = FUSED ? S.in.n : 0, main_ppt = FUSED ? S.in.ppt : 0;
  EvalConsts c;
  c.pts = in.pts, c.img = in.img, c.n = in.n, c.w = in.w, c.h = in.h, c.ppt = in.ppt;
  c.fx = in.fx, c.fy = in.fy, c.cx = in.cx, c.cy = in.cy, c.huber = in.huber;
#pragma unroll
  for (int i = 0; i < 9; i++) c.Ki[i] = in.Ki[i], c.M[i] = in.M[i];
  c.t[0] = in.t[0], c.t[1] = in.t[1], c.t[2] = in.t[2];
  c.aff0 = in.aff0, c.aff1 = in.aff1, c.b0 = in.b0, c.scale = in.scale, c.cutoff = in.cutoff, c.max_energy = in.max_energy;
  c.residual_only = in.residual_only;
  eval_consts_defaults(c);
  asm volatile("" ::"s"(s_status), "s"(s_lvl), "s"(s_kind), "s"(s_spec_valid), "s"(c.pts), "s"(c.img), "s"(c.n), "s"(c.w), "s"(c.h),
               "s"(c.residual_only));
  if (FUSED) asm volatile("" ::"s"(trk_ptr), "s"(main_n), "s"(main_ppt));
  asm volatile("" ::"s"(c.fx), "s"(c.fy), "s"(c.cx), "s"(c.cy), "s"(c.huber), "s"(c.t[0]), "s"(c.t[1]), "s"(c.t[2]), "s"(c.aff0),
               "s"(c.aff1), "s"(c.b0), "s"(c.scale), "s"(c.cutoff), "s"(c.max_energy));
  asm volatile("" ::"s"(c.M[0]), "s"(c.M[1]), "s"(c.M[2]), "s"(c.M[3]), "s"(c.M[4]), "s"(c.M[5]), "s"(c.M[6]), "s"(c.M[7]), "s"(c.M[8]));
  if (LVL0) asm volatile("" ::"s"(c.Ki[0]), "s"(c.Ki[1]), "s"(c.Ki[2]), "s"(c.Ki[3]), "s"(c.Ki[4]), "s"(c.Ki[5]), "s"(c.Ki[6]), "s"(c.Ki[7]), "s"(c.Ki[8]));
  __shared__ int arrive; // tickets of the waves' row sums (eval_chunk_impl)
  if (!FUSED) {
    if (threadIdx.x == 0) arrive = 0;
    __syncthreads(); // (the waves of a workgroup start together and the loads above are in flight: costs nothing)
  }
  if (s_status != ST_RUNNING || s_lvl != lvl || s_kind != MODE) return;
  const bool have_eval = !cand || s_spec_valid != 0;
  if (!FUSED && !have_eval) return;
  if (ROSEL == 1 && c.residual_only) return;
  if (ROSEL == 2 && !c.residual_only) return;
  const int n = c.n;
  const int P = c.ppt;
  const int nchunks = (n + kThreads * P - 1) / (kThreads * P);
  // XCD-aware chunk mapping: workgroup b is dispatched to XCD b % 8, so give each XCD a
  // contiguous band of the template (and therefore of the target rows it gathers from).  Levels of fewer than 8 chunks get
  // exactly as many workgroups as chunks (a launch over hundreds of problems is bound by the workgroup dispatch rate there:
  // rounding 2 chunks up to 8 made the coarsest level's launches 2.5x longer).
  const int per_xcd = gridDim.x >> 3;
  const int chunk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3));
  float *partials_prob = partials + (size_t)prob * partial_stride;
  const int spec_off = partial_stride >> 1; // the speculative candidate's partials: second half of the problem's block
  if (chunk < nchunks && have_eval) {
    __shared__ float red[16][kNumSlots];
    int *const arr = FUSED ? nullptr : &arrive;
    float *const out = partials_prob + (cand ? spec_off : 0) + (size_t)chunk * kPartialStride;
    if (ROSEL == 2)
      eval_chunk_impl<MODE, LVL0, true>(c, chunk, threadIdx.x, true, red, out, arr);
    else
      eval_chunk<MODE, LVL0>(c, chunk, threadIdx.x, true, red, out, arr);
    if (FUSED && threadIdx.x < 64) xwg_release(); // wave 0 stored the partial: performed at device scope before the ticket
  } else if (!FUSED) {
    return;
  }
  if (FUSED) {
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = atomicAdd(&tickets[prob], 1);
      last = t == (int)gridDim.x * (spec_nprob > 0 ? 2 : 1) - 1; // both candidates' rows arrive at the same counter
      if (last) tickets[prob] = 0; // for the next launch
    }
    __syncthreads();
    if (!last) return;
    xwg_acquire(); // the other workgroups' partials were announced by their tickets
    __shared__ LmShared sh;
    __shared__ LmSpecShared sps;
    // (the pending evaluation's point count came with this workgroup's own inputs, the tracker pointer with them: the step asks for
    // state, descriptor and partials in ONE round trip -- one frame in flight is a chain of these launches)
    LmPre pre;
    pre.n_lvl = main_n, pre.ppt_lvl = main_ppt, pre.have_sv = false;
    lm_step_block(MODE, lvl, prob, trk_ptr, const_cast<LMState &>(states[prob]), partials_prob, sh, threadIdx.x,
                  status_out, spec_nprob > 0 ? &sps : nullptr, spec_off, &pre);
  }
}

template <int MODE>
static void launch_eval_ml(hipStream_t s, int lvl, dim3 grid, const TrackerDev *const *trackers, const LMState *states,
                           float *partials, int partial_stride, int *tickets, int *status_out, int spec_nprob, bool split_ro, const int *rowmap) {
  if (tickets) { // fused LM step (never level 0: its kernel stays a pure evaluation)
    hipLaunchKernelGGL((eval_kernel<MODE, false, true>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
  } else if (lvl == 0 && split_ro) {
    hipLaunchKernelGGL((eval_kernel<MODE, true, false, 1>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
    hipLaunchKernelGGL((eval_kernel<MODE, true, false, 2>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
  } else if (lvl == 0)
    hipLaunchKernelGGL((eval_kernel<MODE, true, false>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
  else if (split_ro) {
    hipLaunchKernelGGL((eval_kernel<MODE, false, false, 1>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
    hipLaunchKernelGGL((eval_kernel<MODE, false, false, 2>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
  } else
    hipLaunchKernelGGL((eval_kernel<MODE, false, false>), grid, dim3(kThreads), 0, s, trackers, states, partials,
                       partial_stride, lvl, tickets, status_out, spec_nprob, rowmap);
}

// tickets != nullptr (levels >= 1 only): the kernel also performs the LM step (no lm_kernel launch needed)
void launch_eval(hipStream_t s, int mode, int lvl, int grid_x, int nprob,
                 const TrackerDev *const *trackers, const LMState *states, float *partials,
                 int partial_stride, int *tickets, int *status_out, bool spec, bool split_ro, const int *rowmap) {
  dim3 grid(grid_x, spec ? 2 * nprob : nprob);
  const int spec_nprob = spec ? nprob : 0;
  if (lvl == 0) tickets = nullptr;
  if (mode == 0)
    launch_eval_ml<0>(s, lvl, grid, trackers, states, partials, partial_stride, tickets, status_out, spec_nprob, split_ro, rowmap);
  else if (mode == 2)
    launch_eval_ml<2>(s, lvl, grid, trackers, states, partials, partial_stride, tickets, status_out, spec_nprob, split_ro, rowmap);
  else
    launch_eval_ml<1>(s, lvl, grid, trackers, states, partials, partial_stride, tickets, status_out, spec_nprob, split_ro, rowmap);
}

// one thread: the state machine of a new problem at its coarsest level (:451-474 / :854-872), first evaluation staged
__device__ __forceinline__ void lm_start_problem(const TrackerDev &T, LMState &S, int mode, const StartInfo &I) {
  S.is_scale = mode;
  S.coarsest = I.coarsest;
  S.have_repeated = 0;
  S.lambda = 0.01f;
  S.inc_norm = 0;
  S.inc_f = 0;
  for (int i = 0; i < 7; i++) S.cur[i] = I.pose[i];
  S.aff_cur[0] = I.aff[0];
  S.aff_cur[1] = I.aff[1];
  S.scale_cur = I.scale;
  for (int i = 0; i < DSM_MAX_LEVELS; i++) {
    S.last_residuals[i] = __builtin_nan(""); // :459 / :860
    S.last_inners[i] = 0;
    S.min_res[i] = I.min_res[i];
    S.evals[i] = 0;
    S.evals_ro[i] = 0;
    S.rounds[i] = 0;
  }
  S.spec_valid = 0;
  S.ticks = 0;
  S.n0 = T.lv[0].n;
  S.flow[0] = S.flow[1] = S.flow[2] = 1000; // :460
  S.status = ST_RUNNING;
  begin_level(T, S, I.coarsest);
}

__global__ __launch_bounds__(kLmThreads) void lm_kernel(int mode, int op, int lvl,
                                                        const TrackerDev *const *__restrict__ trackers,
                                                        LMState *__restrict__ states,
                                                        const float *__restrict__ partials, int partial_stride,
                                                        const StartInfo *__restrict__ start,
                                                        SingleOut *__restrict__ single_out,
                                                        int *__restrict__ status_out, int spec, const int *__restrict__ rowmap) {
  const int prob = rowmap ? rowmap[blockIdx.x] : blockIdx.x; // (compact launch: see eval_kernel)
  const bool pose_like = mode != 1; // 0: frame tracking, 2: loop-closure pose -- same 8-DoF LM; 1: stereo scale
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const TrackerDev &T = *trackers[prob];
  LMState &S = states[prob];
  __shared__ LmShared sh;
  __shared__ LmSpecShared sps;

  if (op == LM_OP_START) {
    if (tid == 0) {
      lm_start_problem(T, S, mode, start[prob]);
      if (status_out) {
        status_out[2 * prob] = S.status;
        status_out[2 * prob + 1] = S.lvl;
      }
    }
    return;
  }
  if (op == LM_OP_SINGLE_PREP) {
    if (tid == 0) {
      const StartInfo &I = start[prob];
      S.is_scale = mode;
      S.status = ST_RUNNING;
      S.lvl = I.lvl;
      S.spec_valid = 0;
      make_eval_any(T, S, mode, I.lvl, I.pose, I.aff, I.scale, I.cutoff);
    }
    return;
  }

  // ---- LM_OP_STEP / LM_OP_SINGLE_FINISH ----
  // (one round trip: status, level, kind, the pending evaluation's point count, the tracker pointer and this thread's block of the state)
  LmPre pre;
  const int s_status = S.status, s_lvl = S.lvl, s_kind = S.is_scale;
  pre.n_lvl = S.in.n, pre.ppt_lvl = S.in.ppt, pre.have_sv = true;
  const TrackerDev *Tg = trackers[prob];
  pre.sv = tid < kLmS16 ? ((const uint4 *)&S)[tid] : uint4{0, 0, 0, 0};
  const bool active = (s_status == ST_RUNNING && s_lvl == lvl && s_kind == mode);
  if (!active) { // block-uniform
    if (tid == 0 && status_out) {
      status_out[2 * prob] = s_status;
      status_out[2 * prob + 1] = s_lvl;
    }
    return;
  }
  if (op == LM_OP_STEP) {
    lm_step_block(mode, lvl, prob, Tg, S, partials + (size_t)prob * partial_stride, sh, tid, status_out, spec ? &sps : nullptr, partial_stride >> 1, &pre);
    return;
  }

  // LM_OP_SINGLE_FINISH: reduced sums -> rs, H, b of one evaluation (dsm_tracker_calc_res_*)
  reduce_partials_groups(partials + (size_t)prob * partial_stride, chunks_of(S.in.n, S.in.ppt), tid, sh.red);
  __syncthreads();
  if (tid >= 64) return;
  reduce_partials_final(lane, sh.red);
  const double *sums = sh.red.sums;
  const long long *isums = sh.red.isums;
  double rs[6];
  build_rs(sums, isums, S.in.max_energy, rs);
  const int n_warped = (int)isums[2];
  const int n4 = (n_warped + 3) & ~3;
  const int r = lane >> 3, c = lane & 7;
  SingleOut &O = single_out[prob];
  if (pose_like) {
    O.H[lane] = build_H_elem(T.p, sums, n4, r, c);
    if (lane < 8) O.b[lane] = build_b_elem(T.p, sums, n4, lane);
  }
  if (lane == 0) {
    for (int i = 0; i < 6; i++) O.rs[i] = rs[i];
    O.n_warped = n4;
    if (pose_like) {
      O.Hs = O.bs = 0;
    } else {
      O.Hs = (float)sums[0] * (1.0f / n4); // :1003-1004
      O.bs = (float)sums[1] * (1.0f / n4);
    }
    S.status = ST_IDLE;
  }
}

// ------------------------------------------------------------------------------------------
// coarse_kernel: the whole LM loop of the small pyramid levels inside ONE launch, on LDS-resident data.
// One 512-thread workgroup per problem.  On entering a level it copies the level's intensity plane (and the template, when
// both fit) into LDS with coalesced 16-byte loads; every LM iteration of the level then evaluates its chunks (two at a
// time, each by a 256-thread group running exactly the code of eval_kernel, taps by ds_read instead of global gathers),
// reduces the chunk partials (same fixed order, partials in LDS) and steps the state machine -- state and tracker
// descriptor staged in LDS for the whole loop -- until the problem reaches a level whose plane does not fit (left to the
// launch-per-step path) or terminates.  With `spec` the speculative second candidate of dsm_params.speculate is evaluated
// in the same round and its proposal staged by wave 1 beside wave 0's, as in lm_step_block.  Same arithmetic, same chunk
// geometry, same summation order: bit-identical results to the launch-per-step path; what disappears is every launch
// boundary, every state round trip through global memory and every gather miss of the levels that need the most LM
// iterations and have the least work per iteration.
// Two workgroups per CU (LDS: ~39 KB static + the arena; registers: four waves per SIMD).
// ------------------------------------------------------------------------------------------
#ifndef DSM_COARSE_THREADS
#define DSM_COARSE_THREADS 512
#endif
#ifndef DSM_COARSE_WAVES
#define DSM_COARSE_WAVES 4
#endif
constexpr int kCoarseThreads = DSM_COARSE_THREADS;
constexpr int kCoarseGroups = kCoarseThreads / 256;
constexpr int kCoarseMaxChunks = 20;      // chunks of one evaluation whose partials the kernel keeps in LDS
constexpr int kCoarseArenaFloats = 9216;  // 36 KB: planes up to 120 x 67 (level 4 of 1920 x 1080), 156 x 48 (level 3 of 1248 x 384)

// a level runs in coarse_kernel iff its target plane fits the arena and its chunk partials fit the LDS block (host and device)
__host__ __device__ inline bool coarse_level_ok(int w, int h, int nchunks, int arena_floats) {
  return ((w * h + 3) & ~3) <= arena_floats && nchunks <= kCoarseMaxChunks;
}

template <int MODE, bool LVL0>
__device__ __forceinline__ void eval_chunk_lds(const EvalConsts &c, int chunk, int tid, bool active, float (*red)[kNumSlots], float *out) {
  // (all branches are workgroup-uniform: every thread group of a round evaluates the same candidate)
  if (LVL0) { // level 0 of a tiny image: flow indicators; the template stays in global memory
    if (c.residual_only)
      eval_chunk_impl<MODE, true, true, true, false>(c, chunk, tid, active, red, out);
    else
      eval_chunk_impl<MODE, true, false, true, false>(c, chunk, tid, active, red, out);
  } else if (c.lds_pts) {
    if (c.residual_only)
      eval_chunk_impl<MODE, false, true, true, true>(c, chunk, tid, active, red, out);
    else
      eval_chunk_impl<MODE, false, false, true, true>(c, chunk, tid, active, red, out);
  } else {
    if (c.residual_only)
      eval_chunk_impl<MODE, false, true, true, false>(c, chunk, tid, active, red, out);
    else
      eval_chunk_impl<MODE, false, false, true, false>(c, chunk, tid, active, red, out);
  }
}

// wave-uniform evaluation inputs: LDS -> SGPRs
__device__ __forceinline__ void eval_consts_from_lds(const EvalIn &in, EvalConsts &c) {
  auto rf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const unsigned long long pp = (unsigned long long)in.pts, ip = (unsigned long long)in.img;
  c.pts = (const float4 *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pp >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)pp));
  c.img = (const float *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ip >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)ip));
  c.n = __builtin_amdgcn_readfirstlane(in.n);
  c.ppt = __builtin_amdgcn_readfirstlane(in.ppt);
  c.w = __builtin_amdgcn_readfirstlane(in.w);
  c.h = __builtin_amdgcn_readfirstlane(in.h);
  c.fx = rf(in.fx), c.fy = rf(in.fy), c.cx = rf(in.cx), c.cy = rf(in.cy), c.huber = rf(in.huber);
#pragma unroll
  for (int i = 0; i < 9; i++) c.Ki[i] = rf(in.Ki[i]), c.M[i] = rf(in.M[i]);
  c.t[0] = rf(in.t[0]), c.t[1] = rf(in.t[1]), c.t[2] = rf(in.t[2]);
  c.aff0 = rf(in.aff0), c.aff1 = rf(in.aff1), c.b0 = rf(in.b0), c.scale = rf(in.scale);
  c.cutoff = rf(in.cutoff), c.max_energy = rf(in.max_energy);
  c.residual_only = __builtin_amdgcn_readfirstlane(in.residual_only);
  eval_consts_defaults(c);
}

template <int MODE>
__global__ __launch_bounds__(kCoarseThreads) __attribute__((amdgpu_waves_per_eu(DSM_COARSE_WAVES, DSM_COARSE_WAVES))) void coarse_kernel(
    const TrackerDev *const *__restrict__ trackers, LMState *__restrict__ states, int *__restrict__ status_out, int arena_floats, int spec) {
  extern __shared__ __attribute__((aligned(16))) float arena[];
  const int prob = blockIdx.x;
  const int tid = threadIdx.x;
  LMState &S = states[prob];
  __shared__ LmShared sh;
  __shared__ LmSpecShared sps;
  __shared__ float red[kCoarseGroups][16][kNumSlots];
  __shared__ __attribute__((aligned(16))) float part[2][kCoarseMaxChunks][kPartialStride];
  if (!(S.status == ST_RUNNING && S.is_scale == MODE)) return; // workgroup-uniform
  stage_in(sh.st, &S, tid, kCoarseThreads);
  stage_in(sh.trk, trackers[prob], tid, kCoarseThreads);
  __syncthreads();
  const unsigned arena_lds = (unsigned)(unsigned long long)(DSM_LDS float *)arena;
  const int vb = tid >> 8, t256 = tid & 255;
  bool stepped = false;
  int staged_lvl = -1;
  unsigned lds_pts = 0;
  for (;;) {
    const EvalIn &in = sh.st.in;
    const int status = sh.st.status, lvl = __builtin_amdgcn_readfirstlane(sh.st.lvl);
    EvalConsts c;
    eval_consts_from_lds(in, c);
    if (status != ST_RUNNING || !coarse_level_ok(c.w, c.h, chunks_of(c.n, c.ppt), arena_floats)) break; // workgroup-uniform
    stepped = true;
    if (lvl != staged_lvl) { // entering a level: its plane (and the template, when both fit) -> LDS
      const int px4 = (c.w * c.h + 3) >> 2; // (planes carry four rows of slack: reading up to three floats past w*h is safe)
      const uint4 *src = (const uint4 *)c.img;
      uint4 *dst = (uint4 *)arena;
      for (int i = tid; i < px4; i += kCoarseThreads) dst[i] = src[i];
      lds_pts = 0;
      if (4 * px4 + 4 * c.n <= arena_floats) {
        const fvec4 *ps = (const fvec4 *)c.pts;
        fvec4 *pd = (fvec4 *)(arena + 4 * px4);
        for (int i = tid; i < c.n; i += kCoarseThreads) pd[i] = ps[i];
        lds_pts = arena_lds + 16u * (unsigned)px4;
      }
      staged_lvl = lvl;
      __syncthreads();
    }
    c.lds_img = arena_lds;
    c.lds_pts = lds_pts;
    const int nch = chunks_of(c.n, c.ppt);
    const bool have_spec = spec && sh.st.spec_valid != 0; // workgroup-uniform
    for (int cand = 0; cand < (have_spec ? 2 : 1); cand++) {
      if (cand == 1) {
        eval_consts_from_lds(sh.st.spec_in, c);
        c.lds_img = arena_lds;
        c.lds_pts = lds_pts;
      }
      for (int c0 = 0; c0 < nch; c0 += kCoarseGroups) {
        const int chunk = c0 + vb;
        const bool active = chunk < nch;
        if (lvl == 0)
          eval_chunk_lds<MODE, true>(c, chunk, t256, active, red[vb], part[cand][active ? chunk : 0]);
        else
          eval_chunk_lds<MODE, false>(c, chunk, t256, active, red[vb], part[cand][active ? chunk : 0]);
        __syncthreads(); // red[] is reused by the next round
      }
    }
    if (vb == 0) reduce_partials_groups(&part[0][0][0], nch, t256, sh.red);
    if (spec && vb == (kCoarseGroups > 1 ? 1 : 0)) // (garbage where no speculative candidate was evaluated: never looked at then)
      reduce_partials_groups(&part[1][0][0], nch, t256, sps.red);
    if (tid == 0 && spec) sps.cmd = 0, sps.done = 0;
    __syncthreads();
    if (tid < 64) {
      reduce_partials_final(tid, sh.red);
      if (spec) reduce_partials_final(tid, sps.red);
      lm_step_wave0(MODE, lvl, sh.trk, sh.st, sh, tid, spec ? &sps : nullptr);
    } else if (spec && tid < 128) {
      lm_spec_wave1(MODE, sh.trk, sh.st, sps, tid - 64);
    }
    __syncthreads(); // the state (status, level, next evaluation inputs) is read by all waves
  }
  if (stepped) stage_out(&S, sh.st, tid, kCoarseThreads);
  if (tid == 0 && status_out) {
    status_out[2 * prob] = sh.st.status;
    status_out[2 * prob + 1] = sh.st.lvl;
  }
}

void launch_coarse(hipStream_t s, int mode, int nprob, const TrackerDev *const *trackers, LMState *states,
                   int *status_out, int max_px, bool spec) {
  if (max_px > kCoarseArenaFloats) max_px = kCoarseArenaFloats;
  {
    // never ask for more LDS than the device gives a workgroup (gfx950: 160 KB; the kernel's static part is ~39 KB)
    static int lds_limit = 0;
    if (!lds_limit) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
        lds_limit = v;
      else
        lds_limit = 64 * 1024;
    }
    const int room = (lds_limit - 40 * 1024) / (int)sizeof(float);
    if (max_px > room) max_px = room > 0 ? room : 0;
  }
  const int arena_floats = (max_px + 3) & ~3;
  dim3 grid(nprob), block(kCoarseThreads);
  const size_t dyn = sizeof(float) * (size_t)arena_floats;
  if (mode == 0)
    hipLaunchKernelGGL((coarse_kernel<0>), grid, block, dyn, s, trackers, states, status_out, arena_floats, spec ? 1 : 0);
  else if (mode == 1)
    hipLaunchKernelGGL((coarse_kernel<1>), grid, block, dyn, s, trackers, states, status_out, arena_floats, spec ? 1 : 0);
  else
    hipLaunchKernelGGL((coarse_kernel<2>), grid, block, dyn, s, trackers, states, status_out, arena_floats, spec ? 1 : 0);
}
bool coarse_level_fits(int w, int h, int n, int geom, int max_px) {
  if (max_px > kCoarseArenaFloats) max_px = kCoarseArenaFloats;
  return coarse_level_ok(w, h, num_chunks(n, geom), (max_px + 3) & ~3);
}

// ------------------------------------------------------------------------------------------
// queue_kernel: the whole call in ONE launch of persistent workgroups pulling (problem, chunk) items from a
// device-side queue.  The workgroup that completes a problem's evaluation (arrival ticket) performs its LM step
// and pushes the chunks of the problem's next evaluation, so every problem advances at its own pace: no
// lock-step launches sized for the slowest problem, no idle workgroups, LM steps overlapped with other
// problems' evaluations.  Same chunks, same partials, same reduction order as the launch-per-step path:
// bit-identical results.  Everything one workgroup writes and another reads inside the launch (queue
// items, partials, LMState, tickets) uses device-scope accesses; the XCD L2s are not mutually coherent.
// The grid is sized for co-residency (occupancy query) because that is what performs; correctness does not depend
// on it: the queue is seeded by its own launch and a workgroup only ever waits for an item that some running
// workgroup will publish.  One queue kernel per context at a time (the header and ring belong to the context).
// Waits are bounded and raise q->error instead of hanging.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned q_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// whole wave: append `nitems` chunk items of problem `prob`.  The caller has made the problem's new state visible.
// A ring slot is reused only after its previous item has been READ (the reader stores 0 into it): a workgroup that
// is delayed between taking its ticket and reading its slot can therefore never find the slot overwritten, however
// far the other problems have advanced meanwhile.
__device__ __forceinline__ void queue_push(WorkQueue *q, unsigned long long *items, unsigned qmask, int prob, int nitems, int lane) {
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(&q->tail, (unsigned)nitems);
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  for (int i = lane; i < nitems; i += 64) {
    const unsigned idx = base + (unsigned)i;
    unsigned long long *slot = &items[idx & qmask];
    for (unsigned spins = 0; __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull; spins++) {
      if (spins > (1u << 22)) { // never hang the GPU
        __hip_atomic_store(&q->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(slot, ((unsigned long long)(idx + 1u) << 32) | ((unsigned)prob << kQueueChunkBits) | (unsigned)i,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The first evaluation of every problem enters the queue in a launch of its own (one wave per problem, after
// LM_OP_START): the persistent kernel's progress then never depends on which of its workgroups are resident.
__global__ __launch_bounds__(256) void queue_seed_kernel(int mode, const LMState *__restrict__ states, WorkQueue *__restrict__ q,
                                                         unsigned long long *__restrict__ items, unsigned qmask, int nprob) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= nprob) return;
  const LMState &S0 = states[p];
  if (S0.status == ST_RUNNING && S0.is_scale == mode) {
    const int nch = chunks_of(S0.in.n, S0.in.ppt);
    queue_push(q, items, qmask, p, nch > 0 ? nch : 1, lane);
  } else if (lane == 0) {
    atomicAdd(&q->done, 1u);
  }
}

template <int MODE>
__global__ __launch_bounds__(kThreads, 4) void queue_kernel(const TrackerDev *const *__restrict__ trackers, LMState *__restrict__ states,
                                                         float *__restrict__ partials, int partial_stride, int *__restrict__ tickets,
                                                         WorkQueue *__restrict__ q, unsigned long long *__restrict__ items, unsigned qmask,
                                                         int nprob) {
  __shared__ LmShared sh;
  __shared__ float red[16][kNumSlots];
  __shared__ __attribute__((aligned(16))) EvalIn s_in;
  __shared__ int s_ctl[4]; // problem (-1: leave), chunk, level, "last arrival" flag
  const int tid = threadIdx.x;

  constexpr int kIn16 = sizeof(EvalIn) / 16;
  static_assert(sizeof(EvalIn) % 16 == 0 && kIn16 < 63, "EvalIn is staged with 16-byte copies by wave 0");
  for (;;) {
    // wave 0 takes an item and stages the problem's evaluation inputs; three workgroup barriers per item in all
    if (tid < 64) {
      unsigned it = 0xFFFFFFFFu;
      if (tid == 0) {
        const unsigned t = atomicAdd(&q->head, 1u);
        for (unsigned spins = 0;; spins++) {
          const unsigned long long v = __hip_atomic_load(&items[t & qmask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == t + 1u) {
            it = (unsigned)v;
            __hip_atomic_store(&items[t & qmask], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // slot free again
            break;
          }
          if (q_load(&q->done) >= (unsigned)nprob || q_load((const unsigned *)&q->error)) break;
          if (spins > (1u << 22)) { // ~ a second of polling: something is wrong -- never hang the GPU
            __hip_atomic_store(&q->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
      }
      it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
      if (it != 0xFFFFFFFFu) {
        xwg_acquire(); // the item's publication tag announced the problem's new state
        const LMState &Sp = states[it >> kQueueChunkBits];
        if (tid < kIn16) ((uint4 *)&s_in)[tid] = load16_coherent((const uint4 *)&Sp.in + tid);
        if (tid == kIn16) s_ctl[2] = __hip_atomic_load(&Sp.lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) {
        s_ctl[0] = it == 0xFFFFFFFFu ? -1 : (int)(it >> kQueueChunkBits);
        s_ctl[1] = (int)(it & ((1u << kQueueChunkBits) - 1u));
      }
    }
    __syncthreads();
    const int prob = s_ctl[0], chunk = s_ctl[1];
    if (prob < 0) break; // workgroup-uniform
    LMState &S = states[prob];
    const int lvl = __builtin_amdgcn_readfirstlane(s_ctl[2]);
    EvalConsts c;
    eval_consts_from_lds(s_in, c);
    const int nch = chunks_of(c.n, c.ppt);
    const int nitems = nch > 0 ? nch : 1; // an empty level still needs its LM step
    float *partials_prob = partials + (size_t)prob * partial_stride;
    if (chunk < nch) {
      if (lvl == 0)
        eval_chunk<MODE, true>(c, chunk, tid, true, red, partials_prob + (size_t)chunk * kPartialStride); // (no LDS window here: the kernel's LDS holds the LM
                                                                                                   // step's workspace; a tile-ordered template is gathered -- the same sums)
      else
        eval_chunk<MODE, false>(c, chunk, tid, true, red, partials_prob + (size_t)chunk * kPartialStride);
    }
    if (tid < 64) { // the chunk's partial was stored by threads of wave 0 only (eval_chunk's final sum)
      xwg_release(); // the partial is performed at device scope before the arrival ticket
      if (tid == 0) {
        const int tk = atomicAdd(&tickets[prob], 1);
        const int last = tk == nitems - 1;
        if (last) __hip_atomic_store(&tickets[prob], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ctl[3] = last;
      }
    }
    __syncthreads();
    if (s_ctl[3]) { // workgroup-uniform: the problem's evaluation is complete -> its LM step, then its next evaluation
      xwg_acquire(); // the other chunks' partials were announced by their tickets
      lm_step_block<true>(MODE, lvl, prob, trackers[prob], S, partials_prob, sh, tid, nullptr);
      if (tid < 64) {
        xwg_release(); // new state (and the ticket reset) performed at device scope before the items appear
        if (sh.st.status == ST_RUNNING) {
          const int nn = chunks_of(sh.st.in.n, sh.st.in.ppt);
          queue_push(q, items, qmask, prob, nn > 0 ? nn : 1, tid);
        } else if (tid == 0) {
          atomicAdd(&q->done, 1u);
        }
      }
    }
    // no barrier here: wave 0 rewrites s_ctl[0..2] / s_in only after every wave has passed the barrier above, and all
    // waves read them before the evaluation; the next writes of red[] / sh / s_ctl[3] follow the next item's first barrier
  }
}

int queue_kernel_blocks_per_cu(int mode) {
  int nb = 0;
  if (mode == 0)
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, queue_kernel<0>, kThreads, 0);
  else if (mode == 1)
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, queue_kernel<1>, kThreads, 0);
  else
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, queue_kernel<2>, kThreads, 0);
  return nb;
}

void launch_queue(hipStream_t s, int mode, int nblocks, int nprob, const TrackerDev *const *trackers, LMState *states,
                  float *partials, int partial_stride, int *tickets, WorkQueue *q, unsigned long long *items, unsigned qmask) {
  hipLaunchKernelGGL(queue_seed_kernel, dim3((nprob + 3) / 4), dim3(256), 0, s, mode, states, q, items, qmask, nprob);
#define DSM_QL(M)                                                                                                     \
  hipLaunchKernelGGL((queue_kernel<M>), dim3(nblocks), dim3(kThreads), 0, s, trackers, states, partials, partial_stride, tickets, \
                     q, items, qmask, nprob)
  if (mode == 0)
    DSM_QL(0);
  else if (mode == 1)
    DSM_QL(1);
  else
    DSM_QL(2);
#undef DSM_QL
}

// ------------------------------------------------------------------------------------------
// Tick engine of the streaming form (dsm_stream_*, stream_capi.hip).  A tick advances EVERY resident problem by one LM round,
// whatever level it stands on: tick_eval_kernel evaluates a device-built list of (problem, chunk) items -- all levels mixed in
// one grid, the small levels' chunks filling what the large ones leave --, tick_lm_kernel (one workgroup per slot) steps the
// problems, stages the next tick's items, retires finished problems into a result array and refills their slots from the
// waiting list.  Kernel boundaries order everything: no cross-workgroup protocol, no tickets, no host round trip inside an
// advance.  Same chunks, same partials, same reduction order as every other form: bit-identical results.
// ------------------------------------------------------------------------------------------
// wave 0: the items of the evaluation(s) staged in `St` (the main candidate and, where staged, the speculative one) -- or, where chains
// are on and the evaluation is ONE chunk with no speculative twin, a chain entry: the problem's number behind the list's item slots
// res_base / res_total: item slots this problem reserved EARLY in its step (TickReserve below), to be used or filled with no-ops.
__device__ __forceinline__ void tick_push(const LMState &St, int prob, const TickList &L, TickModeCtl *mc, int lane, int res_base = 0, int res_total = 0) {
  const int nch = chunks_of(St.in.n, St.in.ppt), per_xcd = (nch + 7) >> 3;
  const bool chain = L.chain_cap > 0 && nch == 1 && !St.spec_valid && (L.chain_max_n0 <= 0 || St.n0 <= L.chain_max_n0);
  const int npos = nch < 8 ? nch : 8 * per_xcd;
  int total = chain ? 0 : St.spec_valid ? 2 * npos : npos;
  if (chain && lane == 0) {
    const int idx = atomicAdd(&L.seg->chain[L.buf], 1);
    if (idx < L.chain_cap)
      L.items[L.cap + idx] = (unsigned)prob;
    else
      L.seg->overflow = 1;
    // (the chain books its evaluations itself, round by round)
  }
  int base = res_base;
  if (total > res_total) { // nothing reserved (a reservation is never too short: it is made for the same level, with the twin counted)
    for (int i = lane; i < res_total; i += 64) L.items[res_base + i] = kTickNoop;
    res_total = 0;
    if (lane == 0) {
      base = atomicAdd(&L.seg->count[L.buf], total);
      if (base + total > L.cap) L.seg->overflow = 1;
    }
    base = __builtin_amdgcn_readfirstlane(base);
  }
  if (total > 0 && lane == 0) {
    const int lvl = St.lvl;
    atomicAdd((unsigned long long *)&mc->sched_evals[lvl], 1ull);
    if (St.in.residual_only) atomicAdd((unsigned long long *)&mc->sched_ro[lvl], 1ull);
    atomicAdd((unsigned long long *)&mc->sched_items[lvl], (unsigned long long)total);
  }
  const int fill = total > res_total ? total : res_total; // (total == 0: an empty level -- nothing to evaluate, the LM step runs anyway)
  for (int i = lane; i < fill; i += 64) {
    const bool cand = i >= npos;
    const int p = cand ? i - npos : i;
    // position p runs on XCD (base + p) % 8 (workgroup index = list position + a constant): XCD-contiguous bands of the template, as eval_kernel
    const int chunk = nch < 8 ? p : (p & 7) * per_xcd + (p >> 3);
    const unsigned item = (i < total && chunk < nch) ? (((unsigned)prob << kTickChunkBits) | (unsigned)chunk | (cand ? kTickCand : 0u)) : kTickNoop;
    if (base + i < L.cap) L.items[base + i] = item;
  }
}
// The reservation: handed to the LM step as its `early` hook by tick_lm_kernel.  The step proposes, so the next evaluation is one of the
// level the state stands on: its items (and the speculative twin's, should one be staged) get their place in the list NOW, and the
// atomic's round trip -- 2 k of the 4.9 k cycles "items of the next tick" cost at the end of a step (profiles/r06_tick_stamps.json) --
// runs under the solve.
struct TickReserve {
  TickList L;
  int base, total;
  __device__ __forceinline__ void operator()(const LMState &St, int lane, bool want_spec) {
    const int nch = chunks_of(St.in.n, St.in.ppt), per_xcd = (nch + 7) >> 3;
    const int npos = nch < 8 ? nch : 8 * per_xcd;
    if (L.chain_cap > 0 && nch == 1 && !want_spec && (L.chain_max_n0 <= 0 || St.n0 <= L.chain_max_n0)) return; // (goes to the chain list)
    total = want_spec ? 2 * npos : npos;
    if (total == 0) return;
    int b = 0;
    if (lane == 0) {
      b = atomicAdd(&L.seg->count[L.buf], total);
      if (b + total > L.cap) L.seg->overflow = 1;
    }
    base = b; // (lane 0's value is broadcast where it is used: no wait for the atomic here)
  }
};

// wave 0: entry h of the waiting ring goes into slot `prob`: tracker pointer, ticket, state machine started, first items staged.
// st / trk: LDS scratch of the caller.  stepped: the value of the new state's `stepped` flag (1 when the admission happens inside an
// evaluation launch: the tick's LM launch must leave the newcomer alone).
__device__ __forceinline__ void tick_admit_entry(int mode, long long h, int prob, const TrackerDev **trackers, LMState *states, LMState &st, TrackerDev &trk,
                                                 const TickList &L, TickModeCtl *mc, const TickPending *pending,
                                                 unsigned long long *slot_ticket, int lane, int stepped) {
  const TickPending &Pn = pending[h & (mc->ring - 1)];
  const TrackerDev *tp = Pn.trk;
  stage_in(trk, tp, lane, 64);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane == 0) {
    trackers[prob] = tp;
    slot_ticket[prob] = Pn.ticket;
    lm_start_problem(trk, st, mode, Pn.start);
    st.stepped = stepped;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  stage_out(&states[prob], st, lane, 64);
  tick_push(st, prob, L, mc, lane);
}

// wave 0, inside a tick: the slot whose problem just retired takes the next waiting problem (if any) on the spot.
__device__ __forceinline__ void tick_try_admit(int mode, int prob, const TrackerDev **trackers, LMState *states, LMState &st, TrackerDev &trk,
                                               const TickList &L, TickModeCtl *mc,
                                               const TickPending *pending, unsigned long long *slot_ticket, int lane, int stepped) {
  // pending_head / pending_count are monotonic over the life of the stream (the waiting list is a ring the host appends to while the
  // device consumes): an index is taken by compare-and-swap so that the head never runs past the count -- an overshoot would skip
  // entries the host appends later.  (Few problems retire in one tick, so the swap is rarely contended; the start of an advance,
  // where every free slot wants an entry at once, goes through tick_reserve_kernel instead.)
  long long h = -1;
  if (lane == 0) {
    const long long cnt = __hip_atomic_load(&mc->pending_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long cur = __hip_atomic_load(&mc->pending_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (cur < cnt) {
      const long long seen = (long long)atomicCAS((unsigned long long *)&mc->pending_head, (unsigned long long)cur, (unsigned long long)(cur + 1));
      if (seen == cur) {
        h = cur;
        break;
      }
      cur = seen;
    }
  }
  h = ((long long)__builtin_amdgcn_readfirstlane((int)(h >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)h);
  if (h < 0) return;
  tick_admit_entry(mode, h, prob, trackers, states, st, trk, L, mc, pending, slot_ticket, lane, stepped);
}

// wave 0, after a problem's LM step (state in `st`, already written back): a running problem stages its next evaluation; a terminated
// one writes its result and its slot takes the next waiting problem on the spot
__device__ __forceinline__ void tick_after_step(int mode, int prob, const TrackerDev **trackers, LMState *states, LMState &st, TrackerDev &trk,
                                                const TickList &L, TickModeCtl *mc, const TickPending *pending, TickResult *results,
                                                unsigned long long *slot_ticket, int lane, int stepped, int res_base = 0, int res_total = 0) {
  if (st.status == ST_RUNNING) {
    tick_push(st, prob, L, mc, lane, res_base, res_total);
    return;
  }
  for (int i = lane; i < res_total; i += 64) L.items[res_base + i] = kTickNoop; // (cannot happen: a step that proposes does not terminate)
  if (lane == 0) {
    const long long r = (long long)atomicAdd((unsigned long long *)&mc->retired, 1ull); // monotonic; the host never lets more problems in than the result ring has room for
    TickResult &R = results[r & (mc->ring - 1)];
    const LMState &F = st;
    R.ticket = slot_ticket[prob];
    R.status = F.status;
    R.ticks = F.ticks;
    for (int i = 0; i < 7; i++) R.cur[i] = F.cur[i];
    R.aff_cur[0] = F.aff_cur[0], R.aff_cur[1] = F.aff_cur[1];
    R.flow[0] = F.flow[0], R.flow[1] = F.flow[1], R.flow[2] = F.flow[2];
    R.scale_cur = F.scale_cur;
    for (int l = 0; l < DSM_MAX_LEVELS; l++) {
      R.last_residuals[l] = F.last_residuals[l];
      R.evals[l] = F.evals[l], R.evals_ro[l] = F.evals_ro[l], R.rounds[l] = F.rounds[l];
    }
  }
  tick_try_admit(mode, prob, trackers, states, st, trk, L, mc, pending, slot_ticket, lane, stepped);
}

// Start of an advance, on the context's stream before the stream groups fork (nothing else touches the stream's state then): one
// workgroup hands the entries of the waiting rings to the free slots -- admit_idx[slot] = ring index or -1.  Where fewer problems
// wait than slots are free, the segments (stream groups) of a kind share them in proportion to their free slots.  (Every free
// slot taking its entry by compare-and-swap on the one head word cost 250-350 us per advance: a retry per winner.)
__global__ __launch_bounds__(256) void tick_reserve_kernel(TickReserveArgs a, const LMState *__restrict__ states, TickModeCtl *__restrict__ mcs,
                                                           long long *__restrict__ admit_idx) {
  __shared__ int nfree[kTickMaxSegs], take[kTickMaxSegs], wave_tot[4], running;
  __shared__ long long base[kTickMaxSegs];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < kTickMaxSegs) nfree[tid] = 0;
  __syncthreads();
  for (int si = 0; si < a.nseg; si++) {
    int c = 0;
    for (int j = tid; j < a.seg[si].ns; j += 256) c += states[a.seg[si].i0 + j].status != ST_RUNNING;
    if (c) atomicAdd(&nfree[si], c);
  }
  __syncthreads();
  if (tid == 0) {
    for (int mode = 0; mode < 2; mode++) {
      long long F = 0;
      for (int si = 0; si < a.nseg; si++)
        if (a.seg[si].mode == mode) F += nfree[si];
      const long long head = __hip_atomic_load(&mcs[mode].pending_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long cnt = __hip_atomic_load(&mcs[mode].pending_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long A = cnt - head;
      if (A < 0) A = 0;
      long long given = 0;
      for (int si = 0; si < a.nseg; si++)
        if (a.seg[si].mode == mode) {
          take[si] = F <= A ? nfree[si] : (int)(A * nfree[si] / (F > 0 ? F : 1));
          given += take[si];
        }
      for (int si = 0; si < a.nseg && F > A && given < A; si++) // (the rounding's remainder: one more each, from the first segment on)
        if (a.seg[si].mode == mode && take[si] < nfree[si]) take[si]++, given++;
      long long b = head;
      for (int si = 0; si < a.nseg; si++)
        if (a.seg[si].mode == mode) base[si] = b, b += take[si];
      __hip_atomic_store(&mcs[mode].pending_head, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  for (int si = 0; si < a.nseg; si++) {
    if (tid == 0) running = 0;
    __syncthreads();
    for (int j0 = 0; j0 < a.seg[si].ns; j0 += 256) {
      const int j = j0 + tid;
      const bool fr = j < a.seg[si].ns && states[a.seg[si].i0 + j].status != ST_RUNNING;
      const unsigned long long m = __ballot(fr);
      if (lane == 0) wave_tot[wv] = __popcll(m);
      __syncthreads();
      int r = running + __popcll(m & ((1ull << lane) - 1ull));
      for (int w = 0; w < wv; w++) r += wave_tot[w];
      if (j < a.seg[si].ns) admit_idx[a.seg[si].i0 + j] = fr && r < take[si] ? base[si] + r : -1ll;
      __syncthreads();
      if (tid == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
      __syncthreads();
    }
  }
}

// start of an advance: the free slots take the entries tick_reserve_kernel gave them
__global__ __launch_bounds__(64) void tick_admit_kernel(int mode, const TrackerDev **trackers, LMState *states, TickList L, TickModeCtl *mc,
                                                        const TickPending *pending, unsigned long long *slot_ticket,
                                                        const long long *__restrict__ admit_idx) {
  const int prob = blockIdx.x;
  __shared__ __attribute__((aligned(16))) LMState st;
  __shared__ __attribute__((aligned(16))) TrackerDev trk;
  const long long h = admit_idx[prob];
  if (h < 0) return;
  stage_in(st, &states[prob], threadIdx.x, 64); // (fields the start does not write keep their old bits: none is read)
  tick_admit_entry(mode, h, prob, trackers, states, st, trk, L, mc, pending, slot_ticket, threadIdx.x, 0);
}

// One LM round of a chain (below): the single chunk's partial (LDS) reduced in the fixed order of every other form, wave 0 steps the
// state machine on the LDS copy of the state.  Not inlined: the evaluation loops of tick_eval_kernel keep their registers.
// (LDS objects are handed over as local-address-space pointers: the inlined state machine keeps its ds_ instructions.)
template <int MODE>
__device__ __attribute__((noinline)) void tick_chain_round(DSM_LDS LmShared *shp, const DSM_LDS float *partp, int nch, int lvl, int tid, bool help) {
  LmShared &sh = *(LmShared *)shp;
  const float *part = (const float *)partp;
  reduce_partials_groups(part, nch, tid, sh.red);
  if (tid == 0) sh.help.cmd = 0, sh.help.done = 0;
  __syncthreads();
  if (tid < 64) {
    reduce_partials_final(tid, sh.red);
    lm_step_wave0(MODE, lvl, sh.trk, sh.st, sh, tid, nullptr, MODE != 1 && help);
  } else if (MODE != 1 && help && tid >= 128 && tid < 192) {
    lm_help_wave(sh.trk, sh.st, sh.help, tid - 128);
  }
  __syncthreads(); // the state (status, level, next evaluation inputs) is read by all waves
}
// end of a chain, wave 0: the rounds beyond the first did not cost the problem a tick; state written back with the `stepped` flag up
// (this tick's LM launch leaves the slot alone), then what tick_lm_kernel does after a step
template <int MODE>
__device__ __attribute__((noinline)) void tick_chain_finish(DSM_LDS LmShared *shp, int prob, int rounds, const DSM_LDS TickChainArgs *cap, int lane) {
  LmShared &sh = *(LmShared *)shp;
  const TickChainArgs ca = *(const TickChainArgs *)cap; // (an LDS copy: a by-value struct would go through the stack of EVERY workgroup of the launch)
  if (lane == 0) {
    if (rounds > 1) sh.st.ticks -= rounds - 1;
    sh.st.stepped = 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  stage_out(&ca.states[prob], sh.st, lane, 64);
  tick_after_step(MODE, prob, ca.trackers, ca.states, sh.st, sh.trk, ca.next, ca.mc, ca.pending, ca.results, ca.slot_ticket, lane, 1);
}

// The tick's evaluation launch.  Items [0, count): (problem, chunk) evaluations of all levels mixed, their partials go to memory for
// tick_lm_kernel.  Chain entries [cap, cap + chain): problems whose pending evaluation is ONE chunk (the small pyramid levels -- half of
// a frame's LM rounds on the metric's pyramid, every level of a semi-dense template).  Nothing but this workgroup takes part in such a
// round, so the workgroup that evaluates the chunk also steps the problem -- state and descriptor in LDS, the partial never leaves it --
// and goes on to the NEXT round, for as long as the staged evaluation stays one chunk (at most max_rounds): the rounds that cost a frame
// a tick each (two launches, one trip of the state through memory, a wait for the tick's largest item) cost an evaluation and a step.
// Same chunk, same partial, same reduction order: bit-identical results.  The first workgroups of the grid take the chains (they run
// longest), the others stride over the items.
// CHAIN = false: the launch of a stream that never took in a problem the chains apply to (the host knows: dsm_stream's chain_possible) --
// the items alone, without the chains' code, LDS and stack behind them (the streamed bench of dense templates measured 1.2 % slower
// with them merely present: profiles/r06_ab_lm_opts.log).
template <int MODE, bool CHAIN>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void tick_eval_kernel(const LMState *__restrict__ states,
                                                                                                  float *__restrict__ partials, int partial_stride,
                                                                                                  const unsigned *__restrict__ items,
                                                                                                  TickSegCtl *__restrict__ seg, int buf, int cap, TickChainArgs ca) {
  __shared__ float red[16][kNumSlots];
  const int n_items = ((const DSM_GLOBAL TickSegCtl *)seg)->count[buf];
  const int n_chain = CHAIN ? ((const DSM_GLOBAL TickSegCtl *)seg)->chain[buf] : 0;
  int nc_wg = (int)(gridDim.x >> 1);
  nc_wg = n_chain < nc_wg ? n_chain : nc_wg;
  if (CHAIN && (int)blockIdx.x < nc_wg) {
    __shared__ LmShared sh;
    __shared__ __attribute__((aligned(16))) float part[kPartialStride];
    __shared__ TickChainArgs s_ca;
    const int tid = threadIdx.x;
    if (tid == 0) s_ca = ca;
    for (int ci = blockIdx.x; ci < n_chain; ci += nc_wg) {
      const int prob = (int)((const DSM_GLOBAL unsigned *)items)[cap + ci];
      stage_in(sh.st, &ca.states[prob], tid, kThreads);
      stage_in(sh.trk, ca.trackers[prob], tid, kThreads);
      __syncthreads();
      int rounds = 0;
      for (;;) {
        const int status = __builtin_amdgcn_readfirstlane(sh.st.status), lvl = __builtin_amdgcn_readfirstlane(sh.st.lvl);
        const int spec_valid = __builtin_amdgcn_readfirstlane(sh.st.spec_valid);
        EvalConsts c;
        eval_consts_from_lds(sh.st.in, c);
        const int nch = chunks_of(c.n, c.ppt);
        // one more round while the staged evaluation is one chunk (or none: an empty level), up to the bound: every tick waits for its
        // longest workgroup, and a chain that ran on alone would make every other resident problem wait (measured: the bound of 8)
        if (status != ST_RUNNING || nch > 1 || spec_valid || rounds >= ca.max_rounds) break; // workgroup-uniform
        if (nch == 1) {
          if (lvl == 0)
            eval_chunk<MODE, true>(c, 0, tid, true, red, part);
          else
            eval_chunk<MODE, false, true, 1>(c, 0, tid, true, red, part);
          if (tid == 0) {
            atomicAdd((unsigned long long *)&ca.mc->sched_evals[lvl], 1ull);
            if (c.residual_only) atomicAdd((unsigned long long *)&ca.mc->sched_ro[lvl], 1ull);
            atomicAdd((unsigned long long *)&ca.mc->sched_items[lvl], 1ull);
          }
        }
        __syncthreads();
        tick_chain_round<MODE>((DSM_LDS LmShared *)&sh, (const DSM_LDS float *)part, nch, lvl, tid, (ca.flags & 1) != 0);
        rounds++;
      }
      if (tid < 64) tick_chain_finish<MODE>((DSM_LDS LmShared *)&sh, prob, rounds, (const DSM_LDS TickChainArgs *)&s_ca, tid);
      __syncthreads(); // sh is restaged for the next entry
    }
    return;
  }
  for (int it = (int)blockIdx.x - nc_wg; it < n_items; it += (int)gridDim.x - nc_wg) {
    const unsigned item = ((const DSM_GLOBAL unsigned *)items)[it];
    if (item & kTickNoop) continue; // (workgroup-uniform)
    const bool cand = (item & kTickCand) != 0;
    const int prob = (int)((item & ~(kTickNoop | kTickCand)) >> kTickChunkBits), chunk = (int)(item & ((1u << kTickChunkBits) - 1u));
    const DSM_GLOBAL LMState &S = ((const DSM_GLOBAL LMState *)states)[prob];
    const DSM_GLOBAL EvalIn &in = cand ? S.spec_in : S.in;
    const int lvl = S.lvl;
    EvalConsts c;
    c.pts = in.pts, c.img = in.img, c.n = in.n, c.w = in.w, c.h = in.h, c.ppt = in.ppt;
    c.fx = in.fx, c.fy = in.fy, c.cx = in.cx, c.cy = in.cy, c.huber = in.huber;
#pragma unroll
    for (int i = 0; i < 9; i++) c.Ki[i] = in.Ki[i], c.M[i] = in.M[i];
    c.t[0] = in.t[0], c.t[1] = in.t[1], c.t[2] = in.t[2];
    c.aff0 = in.aff0, c.aff1 = in.aff1, c.b0 = in.b0, c.scale = in.scale, c.cutoff = in.cutoff, c.max_energy = in.max_energy;
    c.residual_only = in.residual_only;
    eval_consts_defaults(c);
    float *const out = partials + (size_t)prob * partial_stride + (cand ? (partial_stride >> 1) : 0) + (size_t)chunk * kPartialStride;
    if (lvl == 0)
      eval_chunk<MODE, true>(c, chunk, threadIdx.x, true, red, out);
    else
      eval_chunk<MODE, false, true, 1>(c, chunk, threadIdx.x, true, red, out); // (the two-point loop here too: this kernel is allocated 128 registers by its level-0 loop anyway)
    __syncthreads(); // red[] is reused by the next item (the arrival-ticket form of eval_kernel measured the same here: profiles/r05_ab_tick_arrival_ticket.log)
  }
}

template <int MODE>
__global__ __launch_bounds__(kLmThreads) void tick_lm_kernel(const TrackerDev **trackers, LMState *__restrict__ states, const float *__restrict__ partials,
                                                             int partial_stride, TickList next, TickModeCtl *__restrict__ mc,
                                                             const TickPending *__restrict__ pending, TickResult *__restrict__ results,
                                                             unsigned long long *__restrict__ slot_ticket, int speculate, int opts) {
  const int prob = blockIdx.x, tid = threadIdx.x;
  LMState &S = states[prob];
  __shared__ LmShared sh;
  __shared__ LmSpecShared sps;
  // the list this tick's evaluation launch consumed (every workgroup of it has finished): empty again for the tick after the next.
  // (Not by the evaluation launch itself: its chains append to the other list while it runs.)
  if (prob == 0 && tid == 0) next.seg->count[next.buf ^ 1] = 0, next.seg->chain[next.buf ^ 1] = 0;
  // ONE round trip for everything the step must know before it can ask for the partials: the slot's status and kind, the level and point
  // count of the pending evaluation, the tracker pointer -- and this thread's block of the state itself
  LmPre pre;
  const int s_status = S.status, s_kind = S.is_scale, lvl = S.lvl, s_stepped = S.stepped;
  pre.n_lvl = S.in.n, pre.ppt_lvl = S.in.ppt, pre.have_sv = true;
  const TrackerDev *Tg = trackers[prob];
  pre.sv = tid < kLmS16 ? ((const uint4 *)&S)[tid] : uint4{0, 0, 0, 0};
  if (s_stepped) { // a chain of this tick's evaluation launch stepped the slot (and staged, retired or refilled it)
    if (tid == 0) S.stepped = 0;
    return;
  }
  if (!(s_status == ST_RUNNING && s_kind == MODE)) return; // a free slot (refilled by the next advance's admit launch)
  // the speculative second candidate (dsm_params.speculate): on the small levels, as in the launch form (a tick's launch is
  // never bound by the doubled rows of a few small problems)
  const bool spec_lvl = speculate >= 2 || (speculate == 1 && pre.n_lvl <= 8192);
  TickReserve rsv{next, 0, 0};
  lm_step_block<false, TickReserve>(MODE, lvl, prob, Tg, S, partials + (size_t)prob * partial_stride, sh, tid, nullptr, spec_lvl ? &sps : nullptr,
                                    partial_stride >> 1, &pre, (opts & 2) ? &rsv : nullptr, (opts & 1) != 0);
  if (tid >= 64) return;
  const int res_base = __builtin_amdgcn_readfirstlane(rsv.base);
  tick_after_step(MODE, prob, trackers, states, sh.st, sh.trk, next, mc, pending, results, slot_ticket, tid, 0, res_base, rsv.total);
}

// chain_kernel: the chain of the tick engine as a launch of its own (dsm_params.persistent_coarse < 0: single calls -- one frame in
// flight).  One workgroup per problem carries it through every level whose evaluation is ONE chunk -- evaluate, reduce, step, on the
// LDS copy of the state -- and hands it back (still RUNNING) at the first level of several chunks: the coarse levels' rounds, each
// a launch of 12-14 us in the launch-per-step form, cost an evaluation and a step.  Same chunk, same partial, same reduction order.
template <int MODE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void chain_kernel(const TrackerDev *const *__restrict__ trackers,
                                                                                              LMState *__restrict__ states, int *__restrict__ status_out) {
  __shared__ float red[16][kNumSlots];
  __shared__ LmShared sh;
  __shared__ __attribute__((aligned(16))) float part[kPartialStride];
  const int prob = blockIdx.x, tid = threadIdx.x;
  LMState &S = states[prob];
  if (!(S.status == ST_RUNNING && S.is_scale == MODE)) return; // workgroup-uniform
  stage_in(sh.st, &S, tid, kThreads);
  stage_in(sh.trk, trackers[prob], tid, kThreads);
  __syncthreads();
  int rounds = 0;
  for (;;) {
    const int status = __builtin_amdgcn_readfirstlane(sh.st.status), lvl = __builtin_amdgcn_readfirstlane(sh.st.lvl);
    const int spec_valid = __builtin_amdgcn_readfirstlane(sh.st.spec_valid);
    EvalConsts c;
    eval_consts_from_lds(sh.st.in, c);
    const int nch = chunks_of(c.n, c.ppt);
    if (status != ST_RUNNING || nch > 1 || spec_valid) break; // workgroup-uniform
    if (nch == 1) {
      if (lvl == 0)
        eval_chunk<MODE, true>(c, 0, tid, true, red, part);
      else
        eval_chunk<MODE, false, true, 1>(c, 0, tid, true, red, part);
    }
    __syncthreads();
    tick_chain_round<MODE>((DSM_LDS LmShared *)&sh, (const DSM_LDS float *)part, nch, lvl, tid, true);
    rounds++;
  }
  if (rounds > 0 && tid < 64) stage_out(&S, sh.st, tid, 64);
  if (tid == 0 && status_out) {
    status_out[2 * prob] = sh.st.status;
    status_out[2 * prob + 1] = sh.st.lvl;
  }
}
void launch_chain(hipStream_t s, int mode, int nprob, const TrackerDev *const *trackers, LMState *states, int *status_out) {
  if (mode == 0)
    hipLaunchKernelGGL((chain_kernel<0>), dim3(nprob), dim3(kThreads), 0, s, trackers, states, status_out);
  else if (mode == 1)
    hipLaunchKernelGGL((chain_kernel<1>), dim3(nprob), dim3(kThreads), 0, s, trackers, states, status_out);
  else
    hipLaunchKernelGGL((chain_kernel<2>), dim3(nprob), dim3(kThreads), 0, s, trackers, states, status_out);
}

void launch_tick_reserve(hipStream_t s, const TickReserveArgs &a, const LMState *states, TickModeCtl *mcs, long long *admit_idx) {
  hipLaunchKernelGGL(tick_reserve_kernel, dim3(1), dim3(256), 0, s, a, states, mcs, admit_idx);
}
void launch_tick_admit(hipStream_t s, int mode, int nslots, const TrackerDev **trackers, LMState *states, const TickList &list, TickModeCtl *mc,
                       const TickPending *pending, unsigned long long *slot_ticket, const long long *admit_idx) {
  hipLaunchKernelGGL(tick_admit_kernel, dim3(nslots), dim3(64), 0, s, mode, trackers, states, list, mc, pending, slot_ticket, admit_idx);
}
void launch_tick_eval(hipStream_t s, int mode, int grid, const LMState *states, float *partials, int partial_stride, const unsigned *items,
                      TickSegCtl *seg, int buf, int cap, const TickChainArgs &chain, bool with_chains) {
  if (grid < 2) grid = 2; // (at least one workgroup for the items beside one for the chains)
  if (mode == 0 && with_chains)
    hipLaunchKernelGGL((tick_eval_kernel<0, true>), dim3(grid), dim3(kThreads), 0, s, states, partials, partial_stride, items, seg, buf, cap, chain);
  else if (mode == 0)
    hipLaunchKernelGGL((tick_eval_kernel<0, false>), dim3(grid), dim3(kThreads), 0, s, states, partials, partial_stride, items, seg, buf, cap, chain);
  else if (with_chains)
    hipLaunchKernelGGL((tick_eval_kernel<1, true>), dim3(grid), dim3(kThreads), 0, s, states, partials, partial_stride, items, seg, buf, cap, chain);
  else
    hipLaunchKernelGGL((tick_eval_kernel<1, false>), dim3(grid), dim3(kThreads), 0, s, states, partials, partial_stride, items, seg, buf, cap, chain);
}
void launch_tick_lm(hipStream_t s, int mode, int nslots, const TrackerDev **trackers, LMState *states, const float *partials, int partial_stride,
                    const TickList &next, TickModeCtl *mc, const TickPending *pending, TickResult *results, unsigned long long *slot_ticket,
                    int speculate, int opts) {
  if (mode == 0)
    hipLaunchKernelGGL((tick_lm_kernel<0>), dim3(nslots), dim3(kLmThreads), 0, s, trackers, states, partials, partial_stride, next, mc, pending,
                       results, slot_ticket, speculate, opts);
  else
    hipLaunchKernelGGL((tick_lm_kernel<1>), dim3(nslots), dim3(kLmThreads), 0, s, trackers, states, partials, partial_stride, next, mc, pending,
                       results, slot_ticket, speculate, opts);
}

int lm_spin_expired() {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_lm_spin_expired), sizeof v) != hipSuccess) return -1;
  return v;
}

void launch_lm(hipStream_t s, int mode, int op, int lvl, int nprob, const TrackerDev *const *trackers,
               LMState *states, const float *partials, int partial_stride, const StartInfo *start,
               SingleOut *single_out, int *status_out, bool spec, const int *rowmap) {
  hipLaunchKernelGGL(lm_kernel, dim3(nprob), dim3(kLmThreads), 0, s, mode, op, lvl, trackers, states, partials,
                     partial_stride, start, single_out, status_out, spec ? 1 : 0, rowmap);
}

// ------------------------------------------------------------------------------------------
// template helpers
// ------------------------------------------------------------------------------------------
__global__ void interleave_kernel(int n, const float *__restrict__ u, const float *__restrict__ v,
                                  const float *__restrict__ id, const float *__restrict__ c,
                                  float4 *__restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = make_float4(u[i], v[i], id[i], c[i]);
}
__global__ void deinterleave_kernel(int n, const float4 *__restrict__ in, float *u, float *v, float *id,
                                    float *c) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = in[i];
    u[i] = p.x;
    v[i] = p.y;
    id[i] = p.z;
    c[i] = p.w;
  }
}
// scaleCoarseDepthL0 (:329-336): lpc_idepth[p] /= scale
__global__ void scale_depth_kernel(int n, float4 *pts, float scale) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pts[i].z = pts[i].z / scale;
}
static inline int grid_for(int n, int block = 256) {
  int g = (n + block - 1) / block;
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

void launch_interleave_template(hipStream_t s, int n, const float *u, const float *v, const float *id,
                                const float *c, float4 *out) {
  if (n > 0) hipLaunchKernelGGL(interleave_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, u, v, id, c, out);
}
void launch_deinterleave_template(hipStream_t s, int n, const float4 *in, float *u, float *v, float *id,
                                  float *c) {
  if (n > 0) hipLaunchKernelGGL(deinterleave_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, in, u, v, id, c);
}
void launch_scale_depth(hipStream_t s, int n, float4 *pts, float scale) {
  if (n > 0) hipLaunchKernelGGL(scale_depth_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, pts, scale);
}
__global__ void scale_depth_levels_kernel(ScaleDepthArgs a, float scale) {
  const int l = blockIdx.y, n = a.n[l];
  float4 *pts = a.pts[l];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pts[i].z = pts[i].z / scale; // (:333, as scale_depth_kernel)
}
void launch_scale_depth_levels(hipStream_t s, const ScaleDepthArgs &a, int max_n, float scale) {
  if (max_n > 0) hipLaunchKernelGGL(scale_depth_levels_kernel, dim3(grid_for(max_n), a.nlevels), dim3(256), 0, s, a, scale);
}

// ------------------------------------------------------------------------------------------
// makeImages (upstream DSO FrameHessian::makeImages; call sites FrontEnd.cpp:605,680)
// The device keeps the INTENSITY plane of every level only; the reference's (I, dx, dy) texels are formed where they are
// consumed (taps_interp) or handed back (dip_export_kernel).
// ------------------------------------------------------------------------------------------
// One launch per pyramid level l: the intensities of level l+1 are the 2x2 means 0.25 * (a + b + c + d) of level l.
// Level 0 also converts the camera image (float or "mono8" bytes, exactly) into the level-0 plane.
template <typename SRC>
__device__ __forceinline__ void pyr_level_body(int wl, int hl, const SRC *__restrict__ src, float *__restrict__ out_l,
                                               float *__restrict__ out_next, int first, int step) {
  const int wn = wl >> 1, hn = hl >> 1;
  const int npx = out_l ? wl * hl : wn * hn;
  for (int idx = first; idx < npx; idx += step) {
    if (out_l) out_l[idx] = (float)src[idx];
    if (out_next && idx < wn * hn) {
      const int x = idx % wn, y = idx / wn;
      const int b = 2 * x + 2 * y * wl;
      out_next[idx] = 0.25f * ((float)src[b] + (float)src[b + 1] + (float)src[b + wl] + (float)src[b + 1 + wl]);
    }
  }
}
// out_l != nullptr: level 0 (src = the camera image); else src = the plane of level l
__global__ void pyr_level_fused_kernel(int wl, int hl, const float *__restrict__ src, float *__restrict__ out_l,
                                       float *__restrict__ out_next) {
  pyr_level_body<float>(wl, hl, src, out_l, out_next, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
// the same for a batch of images (dsm_upload_images): blockIdx.y = image; U8: the level-0 source holds camera bytes
// (main.cpp:216-217 "mono8"), converted exactly
template <bool U8>
__global__ void pyr_level_batched_kernel(int l, int nlevels, int wl, int hl, const PyrJob *__restrict__ jobs) {
  const PyrJob &j = jobs[blockIdx.y];
  float *out_next = l + 1 < nlevels ? j.img[l + 1] : nullptr;
  const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
  if (l == 0) {
    if (U8)
      pyr_level_body<unsigned char>(wl, hl, (const unsigned char *)j.raw, j.img[0], out_next, first, step);
    else
      pyr_level_body<float>(wl, hl, (const float *)j.raw, j.img[0], out_next, first, step);
  } else {
    pyr_level_body<float>(wl, hl, j.img[l], nullptr, out_next, first, step);
  }
}
// The reference's texels from / against an intensity plane.  Gradients: central differences on the flat index for
// idx in [w, w (h - 1)), zero elsewhere and where the difference is not finite (upstream DSO makeImages).
__device__ __forceinline__ void dip_gradients(const float *__restrict__ I, int w, int h, int idx, float &dx, float &dy) {
  dx = 0.f, dy = 0.f;
  if (idx >= w && idx < w * (h - 1)) {
    dx = 0.5f * (I[idx + 1] - I[idx - 1]);
    dy = 0.5f * (I[idx + w] - I[idx - w]);
    if (!__builtin_isfinite(dx)) dx = 0;
    if (!__builtin_isfinite(dy)) dy = 0;
  }
}
// dsm_tracker_get_frame: plane -> (I, dx, dy)
__global__ void dip_export_kernel(int w, int h, const float *__restrict__ I, float *__restrict__ out3) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < w * h; idx += gridDim.x * blockDim.x) {
    float dx, dy;
    dip_gradients(I, w, h, idx, dx, dy);
    out3[3 * idx] = I[idx];
    out3[3 * idx + 1] = dx;
    out3[3 * idx + 2] = dy;
  }
}
// dsm_tracker_upload_frame, step 1: (I, dx, dy) -> plane
__global__ void dip_import_kernel(int npx, const float *__restrict__ in3, float *__restrict__ I) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < npx; idx += gridDim.x * blockDim.x) I[idx] = in3[3 * idx];
}
// step 2: the caller's gradient channels must be what makeImages derives from channel 0 (they are not stored): count
// the texels of the rows makeImages fills whose dx or dy differs (bitwise, or by more than tol * max(1, |g|)) and keep the
// first of them: bad = {count, smallest index}
__global__ void dip_verify_kernel(int w, int h, const float *__restrict__ in3, const float *__restrict__ I, int *__restrict__ bad, float tol) {
  int mine = 0, first = 0x7FFFFFFF;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < w * h; idx += gridDim.x * blockDim.x) {
    if (idx < w || idx >= w * (h - 1)) continue; // rows makeImages leaves untouched are never read by the tracker
    float dx, dy;
    dip_gradients(I, w, h, idx, dx, dy);
    const float gx = in3[3 * idx + 1], gy = in3[3 * idx + 2];
    bool off;
    if (tol > 0.0f)
      off = !(__builtin_fabsf(gx - dx) <= tol * __builtin_fmaxf(1.0f, __builtin_fabsf(dx))) || !(__builtin_fabsf(gy - dy) <= tol * __builtin_fmaxf(1.0f, __builtin_fabsf(dy)));
    else
      off = (__float_as_uint(dx) != __float_as_uint(gx)) || (__float_as_uint(dy) != __float_as_uint(gy));
    if (off) {
      mine++;
      if (idx < first) first = idx;
    }
  }
  if (mine) {
    atomicAdd(bad, mine);
    atomicMin(bad + 1, first);
  }
}
void launch_dip_export(hipStream_t s, int w, int h, const float *plane, float *out3) {
  hipLaunchKernelGGL(dip_export_kernel, dim3(grid_for(w * h)), dim3(256), 0, s, w, h, plane, out3);
}
void launch_dip_import(hipStream_t s, int w, int h, const float *in3, float *plane, int *d_bad, float tol) {
  hipLaunchKernelGGL(dip_import_kernel, dim3(grid_for(w * h)), dim3(256), 0, s, w * h, in3, plane);
  if (d_bad) hipLaunchKernelGGL(dip_verify_kernel, dim3(grid_for(w * h)), dim3(256), 0, s, w, h, in3, plane, d_bad, tol);
}
// descriptors of many trackers in one copy + one launch (after a batched hand-over every tracker's exposure changed)
__global__ void desc_scatter_kernel(const TrackerDev *__restrict__ src, TrackerDev *const *__restrict__ dst) {
  static_assert(sizeof(TrackerDev) % 16 == 0, "TrackerDev is copied in 16-byte units");
  const uint4 *s = (const uint4 *)(src + blockIdx.x);
  uint4 *d = (uint4 *)dst[blockIdx.x];
  for (int i = threadIdx.x; i < (int)(sizeof(TrackerDev) / 16); i += blockDim.x) d[i] = s[i];
}
void launch_desc_scatter(hipStream_t s, int n, const TrackerDev *d_src, TrackerDev *const *d_dst) {
  if (n > 0) hipLaunchKernelGGL(desc_scatter_kernel, dim3(n), dim3(64), 0, s, d_src, d_dst);
}
// Pinned caller images are fetched by the GPU itself (one launch for the whole batch instead of one copy command per
// image): every thread moves UNIT bytes per access straight from host memory.
typedef unsigned copy_uvec4 __attribute__((ext_vector_type(4)));
template <typename UNIT>
__global__ void host_rows_copy_kernel(const PyrJob *__restrict__ jobs, int njobs, int row_units, int rows, size_t pitch_units) {
  // one grid-stride loop over all images: a small grid (asynchronous hand-over) keeps the slow host reads on a few CUs
  const long long per = (long long)row_units * rows, total = per * njobs;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int job = (int)(g / per), idx = (int)(g - (long long)job * per);
    const int r = idx / row_units, c = idx - r * row_units;
    const PyrJob &j = jobs[job];
    ((UNIT *)j.raw)[idx] = __builtin_nontemporal_load((const UNIT *)j.src + (size_t)r * pitch_units + c);
  }
}
void launch_host_rows_copy(hipStream_t s, const PyrJob *d_jobs, int njobs, int row_bytes, int rows, size_t pitch, int unit, int max_blocks) {
  const int row_units = row_bytes / unit;
  long long blocks = ((long long)row_units * rows * njobs + 255) / 256;
  if (blocks > max_blocks) blocks = max_blocks;
  const dim3 grid((unsigned)blocks);
  if (unit == 16)
    hipLaunchKernelGGL(host_rows_copy_kernel<copy_uvec4>, grid, dim3(256), 0, s, d_jobs, njobs, row_units, rows, pitch / 16);
  else if (unit == 4)
    hipLaunchKernelGGL(host_rows_copy_kernel<unsigned>, grid, dim3(256), 0, s, d_jobs, njobs, row_units, rows, pitch / 4);
  else
    hipLaunchKernelGGL(host_rows_copy_kernel<unsigned char>, grid, dim3(256), 0, s, d_jobs, njobs, row_units, rows, pitch);
}
// raw: the level-0 float image; img[l]: the intensity planes of the pyramid levels
void launch_pyramid(hipStream_t s, int w, int h, int nlevels, const float *raw, float *const *img) {
  for (int l = 0; l < nlevels; l++) {
    const int wl = w >> l, hl = h >> l;
    if (l > 0 && l + 1 >= nlevels) break; // the coarsest level has nothing to produce
    hipLaunchKernelGGL(pyr_level_fused_kernel, dim3(grid_for(l == 0 ? wl * hl : (wl >> 1) * (hl >> 1))), dim3(256), 0, s, wl, hl,
                       l == 0 ? raw : img[l], l == 0 ? img[0] : nullptr, l + 1 < nlevels ? img[l + 1] : nullptr);
  }
}
void launch_pyramid_batched(hipStream_t s, int w, int h, int nlevels, const PyrJob *d_jobs, int njobs, bool u8) {
  for (int l = 0; l < nlevels; l++) {
    const int wl = w >> l, hl = h >> l;
    if (l > 0 && l + 1 >= nlevels) break;
    const dim3 grid(grid_for(l == 0 ? wl * hl : (wl >> 1) * (hl >> 1)), njobs);
    if (u8)
      hipLaunchKernelGGL(pyr_level_batched_kernel<true>, grid, dim3(256), 0, s, l, nlevels, wl, hl, d_jobs);
    else
      hipLaunchKernelGGL(pyr_level_batched_kernel<false>, grid, dim3(256), 0, s, l, nlevels, wl, hl, d_jobs);
  }
}

} // namespace dsm

