// xwg_sync.hpp -- the cross-workgroup hand-off primitives of the eval / LM / queue kernels (and of the litmus kernel
// that tests them, diag_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace dsm {

typedef float xwg_fvec4 __attribute__((ext_vector_type(4)));

// Chunk partials are produced by one workgroup and consumed by another (the LM step), possibly on a
// different XCD and -- in the fused eval+LM kernel -- inside the same launch: written and read with
// device-scope accesses (write-through / L2-coherent), so no cache-wide write-back or invalidate is
// ever needed for them.
//
// Ordering between workgroups (arrival tickets, queue items).  Everything one workgroup writes for another inside a
// launch -- chunk partials, the LMState, queue items -- is written with device-scope (sc1, write-through) stores and read
// with device-scope loads, which the non-coherent cache levels do not serve from stale lines.  What the producer still
// owes is that those stores have been PERFORMED before the device-scope atomic that announces them: s_waitcnt vmcnt(0)
// between the two (a workgroup-scope fence alone emits no wait, and the announcing atomic could overtake the stores).
// The textbook form, an agent-scope release fence, emits the same wait plus an L2 write-back (buffer_wbl2 sc1) that has
// nothing to write back here and costs a factor of three on the work-queue kernel (measured on MI355X, 256 dense frames:
// 10.9-11.8 k frames/s with agent-scope release / release+acquire fences against 32.7-33.1 k with this form); an
// agent-scope acquire on the consumer side would invalidate the L2 under the streaming evaluations for the same reason.
// This is the "sc1 payload -> asm vmcnt(0) -> sc1 flag" hand-off of MI355X_MICROARCH.md (handoff-flag row; valid because BOTH
// sides use sc1 accesses for the payload).  The wait is inline assembly on purpose: the guide's compiler-hazard note (ROCm
// 7.2, gfx950) -- the waitcnt-insertion pass drops a builtin wait whose counter it believes empty; inline assembly is
// invisible to it.  tests/test_abi.py::test_ticket_atomics_follow_a_drained_store_queue checks the emitted ISA,
// tests/test_schedule_stress.py::test_cross_xcd_hand_off_litmus the behaviour (10^7 hand-offs across XCDs).
// DSM_TEXTBOOK_FENCES (test builds only, tests/test_schedule_stress.py): the agent-scope release / acquire fences of the
// LLVM memory model instead -- same results, a third of the work-queue kernel's throughput.
#ifdef DSM_TEXTBOOK_FENCES
__device__ __forceinline__ void xwg_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the guide's fix for the dropped wait after buffer_wbl2)
}
__device__ __forceinline__ void xwg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#else
__device__ __forceinline__ void xwg_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // compiler + LDS ordering
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the written-through stores have been acknowledged
}
__device__ __forceinline__ void xwg_acquire() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // later (device-scope) loads are not moved above the announcement
}
#endif
__device__ __forceinline__ void store_partial(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ xwg_fvec4 load_partial4(const float *p) {
  const unsigned long long a = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load((const unsigned long long *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  xwg_fvec4 v;
  v.x = __uint_as_float((unsigned)a), v.y = __uint_as_float((unsigned)(a >> 32));
  v.z = __uint_as_float((unsigned)b), v.w = __uint_as_float((unsigned)(b >> 32));
  return v;
}


} // namespace dsm
