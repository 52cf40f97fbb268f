#!/usr/bin/env python3
"""EXPERIMENT BUILD (round 6), not the product: copies direct_stereo_slam_amd/csrc to scratch/csrc_stamps, adds shader-clock stamps to
the tick engine's kernels and builds scratch/lib_stamps/libdsm_hotpath.so (+ the debug entry point dsm_debug_stamps).
  tick_eval_kernel<pose>: cycles from an item's decode to the end of its evaluation, booked by (level, residual-only)
  tick_lm_kernel<pose>:   cycles between the phase boundaries of an LM workgroup (wave 0, lane 0)
Run:  python tools/experiments/r06_stamps_build.py && gpurun -- 'python tools/experiments/r06_stamps_run.py'
Result of round 6: profiles/r06_tick_stamps.json (before the chunk-table / prefetch changes), profiles/r06_tick_stamps_after.json."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC, DST = os.path.join(ROOT, "direct_stereo_slam_amd", "csrc"), os.path.join(ROOT, "scratch", "csrc_stamps")
shutil.rmtree(DST, ignore_errors=True)
os.makedirs(DST)
for f in os.listdir(SRC):
    if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":
        shutil.copy(os.path.join(SRC, f), DST)
p = os.path.join(DST, "tracker_kernels.hip")
s = open(p).read()


def sub(old, new):
    global s
    assert old in s, old[:80]
    s = s.replace(old, new, 1)


sub('''namespace dsm {


// ------------------------------------------------------------------------------------------
// wave64 helpers''', '''namespace dsm {

// g_stamps[0 .. 23]: tick_eval_kernel, [2 * (2 * lvl + ro)] = sum of workgroup cycles from item start to item end, [+1] = items
// g_stamps[32 .. 43]: tick_lm_kernel phases: sums of cycles between consecutive stamps; [48] = steps with a proposal, [49] = without
__device__ unsigned long long g_stamps[64];
#define DSM_STAMP(scr, i) do { if ((threadIdx.x & 63) == 0) (scr).stamps[i] = clock64(); } while (0)


// ------------------------------------------------------------------------------------------
// wave64 helpers''')
sub('''struct LdltScratch {
  double x[8];''', '''struct LdltScratch {
  unsigned long long stamps[16];
  double x[8];''')
sub('''  double inc[8];
  wave_ldlt_solve8(S.H, S.b, lambda, active, stitch, lane, scr, inc);''', '''  double inc[8];
  if (!spec) DSM_STAMP(scr, 4);
  wave_ldlt_solve8(S.H, S.b, lambda, active, stitch, lane, scr, inc);
  if (!spec) DSM_STAMP(scr, 5);''')
sub('''  make_eval_any(T, S, S.is_scale, S.lvl, cand, aff_cand, 1.0f, T.p.coarse_cutoff_th * S.level_cutoff_repeat, lane == 0, spec, last);
}''', '''  if (!spec) DSM_STAMP(scr, 6);
  make_eval_any(T, S, S.is_scale, S.lvl, cand, aff_cand, 1.0f, T.p.coarse_cutoff_th * S.level_cutoff_repeat, lane == 0, spec, last);
  if (!spec) DSM_STAMP(scr, 7);
}''')
sub('''  const bool pose_like = mode != 1;
  const double *sums = sh.red.sums;''', '''  const bool pose_like = mode != 1;
  DSM_STAMP(sh.ldlt, 3);
  for (int i = 4; i < 8; i++) if (lane == 0) sh.ldlt.stamps[i] = 0;
  const double *sums = sh.red.sums;''')
sub('''  if (lane == 0 && level_done) end_level(T, S);
}''', '''  DSM_STAMP(sh.ldlt, 8);
  if (lane == 0 && level_done) end_level(T, S);
  DSM_STAMP(sh.ldlt, 9);
}''')
sub('''  constexpr int kS16 = kLmS16, kT16 = kLmT16;
  static_assert(kS16 <= kThreads && kT16 <= kThreads, "one 16-byte block per thread");''', '''  constexpr int kS16 = kLmS16, kT16 = kLmT16;
  static_assert(kS16 <= kThreads && kT16 <= kThreads, "one 16-byte block per thread");
  const unsigned long long st_entry = clock64();''')
sub('''  __syncthreads();
  if (sp && tid >= 64 && tid < 128) lm_spec_wave1(mode, sh.trk, sh.st, *sp, tid - 64);
  if (tid >= 64) return; // wave 0 carries on
  const int lane = tid;
  reduce_partials_final(lane, sh.red);''', '''  if (tid == 0) sh.ldlt.stamps[0] = st_entry, sh.ldlt.stamps[1] = clock64();
  __syncthreads();
  if (sp && tid >= 64 && tid < 128) lm_spec_wave1(mode, sh.trk, sh.st, *sp, tid - 64);
  if (tid >= 64) return; // wave 0 carries on
  const int lane = tid;
  DSM_STAMP(sh.ldlt, 2);
  reduce_partials_final(lane, sh.red);''')
sub('''    status_out[2 * prob + 1] = sh.st.lvl;
  }
}

// ------------------------------------------------------------------------------------------
// eval kernel: grid (chunk slots, problems).''', '''    status_out[2 * prob + 1] = sh.st.lvl;
  }
  DSM_STAMP(sh.ldlt, 10);
}
__device__ __forceinline__ void lm_stamps_flush(LmShared &sh, unsigned long long t_kernel_entry) {
  if (threadIdx.x != 0) return;
  unsigned long long *st = sh.ldlt.stamps;
  st[11] = clock64();
  if (st[4] == 0 || st[7] == 0) { atomicAdd(&g_stamps[49], 1ull); return; } // a step without a proposal (level end, cut-off repeat)
  atomicAdd(&g_stamps[32 + 0], st[0] - t_kernel_entry);
  for (int i = 0; i < 11; i++) atomicAdd(&g_stamps[32 + 1 + i], st[i + 1] - st[i]);
  atomicAdd(&g_stamps[48], 1ull);
}

// ------------------------------------------------------------------------------------------
// eval kernel: grid (chunk slots, problems).''')
sub('''  const int prob = blockIdx.x, tid = threadIdx.x;
  LMState &S = states[prob];
  __shared__ LmShared sh;
  __shared__ LmSpecShared sps;
  // ONE round trip for everything''', '''  const int prob = blockIdx.x, tid = threadIdx.x;
  const unsigned long long t_kernel_entry = clock64();
  LMState &S = states[prob];
  __shared__ LmShared sh;
  __shared__ LmSpecShared sps;
  // ONE round trip for everything''')
sub('''  tick_try_admit(MODE, prob, trackers, states, sh.st, sh.trk, items_next, seg, buf_next, cap, mc, pending, slot_ticket, lane);
}''', '''  tick_try_admit(MODE, prob, trackers, states, sh.st, sh.trk, items_next, seg, buf_next, cap, mc, pending, slot_ticket, lane);
  if (MODE == 0) lm_stamps_flush(sh, t_kernel_entry);
}''')
sub('''  if (sh.st.status == ST_RUNNING) {
    tick_push(sh.st, prob, items_next, seg, buf_next, cap, mc, lane);
    return;
  }''', '''  if (sh.st.status == ST_RUNNING) {
    tick_push(sh.st, prob, items_next, seg, buf_next, cap, mc, lane);
    if (MODE == 0) lm_stamps_flush(sh, t_kernel_entry);
    return;
  }''')
sub('''    const DSM_GLOBAL LMState &S = ((const DSM_GLOBAL LMState *)states)[prob];
    const DSM_GLOBAL EvalIn &in = cand ? S.spec_in : S.in;
    const int lvl = S.lvl;
    EvalConsts c;''', '''    const unsigned long long t_item = clock64();
    const DSM_GLOBAL LMState &S = ((const DSM_GLOBAL LMState *)states)[prob];
    const DSM_GLOBAL EvalIn &in = cand ? S.spec_in : S.in;
    const int lvl = S.lvl;
    EvalConsts c;''')
sub('''    __syncthreads(); // red[] is reused by the next item (the arrival-ticket form''', '''    if (MODE == 0 && threadIdx.x == 0) {
      const int k = 2 * (2 * lvl + (c.residual_only ? 1 : 0));
      atomicAdd(&g_stamps[k], clock64() - t_item);
      atomicAdd(&g_stamps[k + 1], 1ull);
    }
    __syncthreads(); // red[] is reused by the next item (the arrival-ticket form''')
s = s.rstrip() + '''
extern "C" int dsm_debug_stamps(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dsm::g_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(dsm::g_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
'''
open(p, "w").write(s)
subprocess.check_call(["make", "-s", "-j", str(os.cpu_count() or 4), "OUT=../lib_stamps"], cwd=DST)
print("built", os.path.join(ROOT, "scratch", "lib_stamps", "libdsm_hotpath.so"))
