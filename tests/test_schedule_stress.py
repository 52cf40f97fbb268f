"""Scheduling choices (work-queue kernel, fused LM step, persistent small-level kernel, stream groups) must never change a result:
repeated runs of mixed batches under every combination, compared bit for bit with the plain two-kernel,
single-stream form.  A race in the ticket / staging protocols would show up here as a sporadic mismatch."""
import numpy as np
import pytest

from _scenes import S, hip_tracker, make_scene

pytestmark = pytest.mark.gpu


def _run(ctx, scs, fuse, ns, coarse, queue=0, spec=0):
    from direct_stereo_slam_amd.tracker import default_params

    p = default_params()
    p.fuse_lm, p.persistent_coarse, p.work_queue, p.speculate = fuse, coarse, queue, spec
    ctx.set_streams(ns)
    out = []
    for parity in (0, 1):  # two batches of equal image size
        idx = [i for i in range(len(scs)) if i % 2 == parity]
        trks = [hip_tracker(ctx, scs[i], p) for i in idx]
        n, nl = len(trks), scs[idx[0]].nl
        r = ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
        e = ctx.optimize_scale_batch(trks, np.full(n, 1.2), nl - 1)
        out.append(tuple(r) + tuple(e))
    return out


def test_every_schedule_gives_identical_results(ctx):
    scs = [make_scene("small" if i % 2 else "medium", seed=200 + i, template="dense" if i % 3 else "sparse", n0=6000)
           for i in range(8)]
    try:
        ref = _run(ctx, scs, 0, 1, 0)
        for it in range(12):
            for fuse, ns, coarse, queue, spec in ((2, 1, 0, 0, 0), (2, 2, 0, 0, 0), (1, 3, 0, 0, 0), (0, 2, 4096, 0, 0), (2, 2, 2048, 0, 0),
                                                  (0, 1, 0, 2, 0), (1, 2, 0, 2, 0),
                                                  # speculative second candidate: two-kernel form, fused form, with the other switches
                                                  (0, 1, 0, 0, 2), (2, 2, 0, 0, 2), (1, 3, 0, 0, 1), (0, 2, 4096, 0, 2), (1, 2, 0, 2, 2)):
                got = _run(ctx, scs, fuse, ns, coarse, queue, spec)
                for g, r in zip(got, ref):
                    for a, b in zip(g, r):
                        assert np.array_equal(a, b, equal_nan=True), (it, fuse, ns, coarse, queue, spec)
    finally:
        ctx.set_streams(1)


def test_speculation_keeps_the_evaluation_counts_and_saves_launches(ctx):
    """dsm_params.speculate: same per-level evaluation counts as the sequential loop (= the oracle's), fewer launches"""
    from _scenes import oracle_tracker
    from direct_stereo_slam_amd.tracker import default_params

    sc = make_scene("medium", seed=31)
    orc = oracle_tracker(sc)
    orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    want = orc.eval_counts()[0][:sc.nl]
    launches = {}
    for spec in (0, 2):
        p = default_params()
        p.speculate, p.fuse_lm = spec, 0
        trk = hip_tracker(ctx, sc, p)
        for _ in range(3):  # let the launch schedule settle
            trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
        st = ctx.stats()
        assert list(st.evals)[:sc.nl] == want
        launches[spec] = sum(st.launches)
    assert launches[2] < launches[0]


def test_last_evaluation_of_a_level_is_residual_only(ctx):
    """The LM loop of a level ends after the evaluation of a step whose increment is below 1e-3 (TrackerAndScaler.cpp:588,
    :937) or at the iteration bound: the reference still runs calcGSSSE* on it and never reads the result.  Here that
    evaluation executes calcRes* alone (EvalIn::residual_only) -- exactly one per level whose loop made a step, counted in
    dsm_stats.evals_residual_only, in every scheduling form, with the same poses, residuals and evaluation counts"""
    from _scenes import oracle_tracker
    from direct_stereo_slam_amd.tracker import default_params

    sc = make_scene("medium", seed=33)
    orc = oracle_tracker(sc)
    good_o, pose_o, aff_o, last_o, _ = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    want = orc.eval_counts()[0][:sc.nl]
    err_o, s_o = orc.optimize_scale(1.0, sc.nl - 1)
    want_s = orc.eval_counts()[0][:sc.nl]
    ref = None
    for fuse, queue, spec, coarse in ((0, 0, 0, 0), (2, 0, 2, 0), (0, 2, 0, 0), (0, 0, 1, 2048)):
        p = default_params()
        p.fuse_lm, p.work_queue, p.speculate, p.persistent_coarse = fuse, queue, spec, coarse
        trk = hip_tracker(ctx, sc, p)
        good, pose, aff, last = trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
        st = ctx.stats()
        assert good == good_o and list(st.evals)[:sc.nl] == want
        np.testing.assert_allclose(pose, pose_o, atol=1e-4)
        np.testing.assert_allclose(last[:sc.nl], last_o[:sc.nl], rtol=1e-4)
        ro = list(st.evals_residual_only)[:sc.nl]
        assert all(r == (1 if e > 1 else 0) for r, e in zip(ro, want)), (ro, want)  # every level's loop made at least one step
        if ref is None:
            ref = (pose, aff, last)
        else:  # scheduling only
            assert np.array_equal(pose, ref[0]) and np.array_equal(aff, ref[1]) and np.array_equal(last, ref[2], equal_nan=True)
        err, s = trk.optimizeScale(1.0, sc.nl - 1)
        st = ctx.stats()
        assert list(st.evals)[:sc.nl] == want_s and abs(s - s_o) < 1e-4
        assert all(r == (1 if e > 1 else 0) for r, e in zip(list(st.evals_residual_only)[:sc.nl], want_s))


def test_track_and_scale_in_one_call_equals_the_two_calls(ctx):
    """dsm_track_and_scale_batch: the scale problems run as a companion segment on their own stream; bit-identical results"""
    from direct_stereo_slam_amd.tracker import default_params

    scs = [make_scene("small", seed=300 + i, template="dense" if i % 2 else "sparse", n0=3000) for i in range(10)]
    p = default_params()
    p.work_queue = 0
    trks = [hip_tracker(ctx, sc, p) for sc in scs]
    n, nl = len(trks), scs[0].nl
    kf = trks[::3]
    try:
        for ns in (1, 2):
            ctx.set_streams(ns)
            for _ in range(4):
                r = ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
                ev_t = list(ctx.stats().evals)
                e, s = ctx.optimize_scale_batch(kf, np.full(len(kf), 1.3), nl - 1)
                ev_s = list(ctx.stats().evals)
                c = ctx.track_and_scale_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1, kf, np.full(len(kf), 1.3))
                for a, b in zip(r, c[:5]):
                    assert np.array_equal(a, b, equal_nan=True)
                assert np.array_equal(e, c[5], equal_nan=True) and np.array_equal(s, c[6])
                assert list(ctx.stats().evals) == ev_t and list(ctx.stats2().evals) == ev_s
        # no keyframe in the step: the call degenerates to dsm_track_batch
        c0 = ctx.track_and_scale_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1, [], [])
        assert np.array_equal(c0[1], r[1]) and len(c0[5]) == 0
    finally:
        ctx.set_streams(1)


def test_cross_xcd_hand_off_litmus(ctx):
    """message passing through the kernels' hand-off primitives (device-scope stores, drained store queue, ticket; device-scope
    loads): 10^7 hand-offs of 256-byte blocks between workgroups on different XCDs under uneven load, every word checked"""
    n, stale = ctx.xwg_litmus(pairs=128, iters=80000)
    assert n >= 10_000_000 and stale == 0


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from direct_stereo_slam_amd.tracker import Context
from test_schedule_stress import _run
from _scenes import make_scene
ctx = Context(0)
scs = [make_scene("small" if i % 2 else "medium", seed=200 + i, template="dense" if i % 3 else "sparse", n0=6000) for i in range(8)]
ref = np.load(sys.argv[2], allow_pickle=True)["ref"]
for it in range(3):
    for fuse, ns, coarse, queue, spec in ((0, 1, 0, 0, 0), (2, 2, 0, 0, 0), (0, 1, 0, 2, 0), (1, 2, 0, 2, 2), (2, 2, 0, 0, 2)):
        got = _run(ctx, scs, fuse, ns, coarse, queue, spec)
        for g, r in zip(got, ref):
            for a, b in zip(g, r):
                assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), (it, fuse, ns, coarse, queue, spec)
n, stale = ctx.xwg_litmus(pairs=64, iters=20000)
assert stale == 0
print("TEXTBOOK-OK", n)
"""


def test_textbook_fences_give_the_same_results(ctx, tmp_path):
    """The hand-off of xwg_sync.hpp (workgroup fence + drained store queue) against the LLVM memory model's own form: a second
    library built with -DDSM_TEXTBOOK_FENCES (agent-scope release / acquire fences) must produce bit-identical results in
    the fused and work-queue forms -- the forms that hand data between workgroups inside a launch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(str(tmp_path), "lib")
    subprocess.run(["make", "-s", "-j", str(os.cpu_count() or 4), "-C", os.path.join(root, "direct_stereo_slam_amd", "csrc"), f"OUT={out}",
                    "EXTRA=-DDSM_TEXTBOOK_FENCES"], check=True, timeout=1200)
    scs = [make_scene("small" if i % 2 else "medium", seed=200 + i, template="dense" if i % 3 else "sparse", n0=6000) for i in range(8)]
    try:
        ref = _run(ctx, scs, 0, 1, 0)
    finally:
        ctx.set_streams(1)
    arr = np.empty(len(ref), dtype=object)
    for i, r in enumerate(ref):
        arr[i] = [np.asarray(x) for x in r]
    np.savez(os.path.join(str(tmp_path), "ref.npz"), ref=arr)
    env = dict(os.environ, DSM_HOTPATH_LIB=os.path.join(out, "libdsm_hotpath.so"), PYTHONPATH=root)
    res = subprocess.run([sys.executable, "-c", _CHILD, os.path.dirname(os.path.abspath(__file__)), os.path.join(str(tmp_path), "ref.npz")],
                         capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert res.returncode == 0 and "TEXTBOOK-OK" in res.stdout, res.stderr[-3000:]


@pytest.mark.parametrize("geometry", [0, 1])
def test_compact_straggler_launches_are_bit_identical(ctx, geometry):
    """dsm_params.compact_tail: after the rounds most problems of a level need, the level's status is read back once and the
    remaining rounds run as compact launches over the stragglers; later passes cover the running problems only.  Same
    evaluations, same steps: bit-identical results, in every combination with the other scheduling switches -- under either
    chunk table (the compact form only sets in for launches of >= 1024 workgroups: 48 of these small scenes reach that under
    the latency table's short chunks, which is where the read-backs are counted)."""
    from direct_stereo_slam_amd.tracker import default_params

    scs = [make_scene("small", seed=400 + i, template="dense" if i % 4 else "sparse", n0=5000, motion_scale=3.0 if i % 5 == 0 else 1.0)
           for i in range(48)]
    poses0 = np.tile(S.IDENTITY_POSE, (len(scs), 1))
    results, polls = {}, {}
    try:
        for compact, ns, fuse, spec in ((0, 1, 0, 0), (1, 1, 0, 0), (1, 2, 1, 1), (0, 2, 1, 1), (1, 3, 2, 2)):
            p = default_params()
            p.compact_tail, p.fuse_lm, p.speculate, p.work_queue, p.chunk_geometry = compact, fuse, spec, 0, geometry
            ctx.set_streams(ns)
            trks = [hip_tracker(ctx, sc, p) for sc in scs]
            out = None
            for rep in range(3):  # the schedule learns from call to call: the compact form sets in at the second
                r = ctx.track_batch(trks, poses0.copy(), np.zeros((len(scs), 2)), scs[0].nl - 1)
                pl = ctx.stats().polls
                e = ctx.optimize_scale_batch(trks[::3], np.full(len(trks[::3]), 1.2), scs[0].nl - 1)
                cur = tuple(np.asarray(x) for x in tuple(r) + tuple(e))
                if out is not None:
                    for a, b in zip(out, cur):
                        assert np.array_equal(a, b, equal_nan=True), ("call-to-call", compact, ns, fuse, spec, rep)
                out = cur
            results[(compact, ns, fuse, spec)], polls[(compact, ns, fuse, spec)] = out, pl
    finally:
        ctx.set_streams(1)
    ref = results[(0, 1, 0, 0)]
    for key, got in results.items():
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True), key
    if geometry == 1:
        assert polls[(1, 1, 0, 0)] > polls[(0, 1, 0, 0)]  # the per-level read-backs really happened
