"""The streaming form of the batched calls (dsm_stream_*, csrc/stream_capi.hip): problems are admitted as slots free up, advance
in passes with a bounded number of rounds per level, are carried over when they need more and retire individually.  None of
that may change a result: every problem's pose, residuals, flags and per-level evaluation counts must equal the batch
calls' bit for bit, whatever the pool size, the rounds per pass or the stream groups."""
import numpy as np
import pytest

from _scenes import S, hip_tracker, make_scene

pytestmark = pytest.mark.gpu


def _batch_reference(ctx, trks, nl, scales):
    n = len(trks)
    good, poses, affs, last, flow = ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
    ev_t = list(ctx.stats().evals)
    err, sc = ctx.optimize_scale_batch(trks, scales, nl - 1)
    ev_s = list(ctx.stats().evals)
    return good, poses, affs, last, flow, err, sc, ev_t, ev_s


def _stream_run(ctx, trks, nl, scales, track_slots, scale_slots, rounds=None, quantile=None, waves=1, engine=0, ticks=0, pipelined=True):  # noqa: PLR0913
    from direct_stereo_slam_amd.tracker import Stream

    n = len(trks)
    st = Stream(ctx, track_slots, scale_slots, engine, ticks)
    if not pipelined:
        st.set_pipelined(False)
    if rounds is not None:
        st.set_rounds(0, rounds)
        st.set_rounds(1, rounds)
    if quantile is not None:
        st.set_quantile(quantile)
    owner = {}
    per = (n + waves - 1) // waves
    passes = 0
    out = []
    for w in range(waves):  # submissions arrive while earlier problems are still resident
        idx = list(range(w * per, min(n, (w + 1) * per)))
        if not idx:
            continue
        tk = st.submit_track([trks[i] for i in idx], np.tile(S.IDENTITY_POSE, (len(idx), 1)), np.zeros((len(idx), 2)), nl - 1)
        ts = st.submit_scale([trks[i] for i in idx], scales[idx], nl - 1)
        for i, a, b in zip(idx, tk, ts):
            owner[a], owner[b] = ("track", i), ("scale", i)
        st.advance()
        passes += 1
        out += st.results()
    while True:
        resident, waiting, _ = st.counts()
        if resident == 0 and waiting == 0:
            break
        assert resident <= track_slots + scale_slots
        st.advance()
        passes += 1
        out += st.results()
        assert passes < 2000
    res = {}
    for r in out:
        assert r.ticket in owner and owner[r.ticket] not in res  # every problem retires exactly once
        res[owner[r.ticket]] = r
    assert len(res) == 2 * n
    sched = st.schedule(0)
    st.close()
    return res, passes, sched


def _check(res, ref, n, nl):
    good, poses, affs, last, flow, err, sc, ev_t, ev_s = ref
    tot_t, tot_s = [0] * 6, [0] * 6
    for i in range(n):
        r = res[("track", i)]
        assert r.kind == 0 and bool(r.good) == bool(good[i])
        assert np.array_equal(np.array(r.pose), poses[i]) and np.array_equal(np.array(r.aff), affs[i])
        assert np.array_equal(np.array(r.last_residuals), last[i], equal_nan=True)
        assert np.array_equal(np.array(r.flow), flow[i])
        q = res[("scale", i)]
        assert q.kind == 1 and np.float32(q.scale) == np.float32(sc[i])
        assert np.array_equal(np.float32(q.err), np.float32(err[i]), equal_nan=True)
        for l in range(6):
            tot_t[l] += r.evals[l]
            tot_s[l] += q.evals[l]
    assert tot_t[:nl] == ev_t[:nl] and tot_s[:nl] == ev_s[:nl]  # the same evaluations, level by level


@pytest.mark.parametrize("streams", [1, 2])
def test_stream_results_equal_the_batch_calls_bit_for_bit(ctx, streams):
    scs = [make_scene("small", seed=700 + i, template="dense" if i % 3 else "sparse", n0=3000) for i in range(20)]
    nl = scs[0].nl
    ctx.set_streams(streams)
    try:
        trks = [hip_tracker(ctx, sc) for sc in scs]
        scales = np.linspace(0.8, 1.3, len(trks)).astype(np.float32)
        ref = _batch_reference(ctx, trks, nl, scales)
        # (a) a pool as large as the job, the learnt schedule; (b) a pool a third of the job: problems wait for slots;
        # (c) two rounds per level and pass: nearly every problem is carried over several passes on several levels;
        # (d) one round per pass, submissions in four waves: the extreme of carrying
        for slots, sslots, rounds, waves in ((20, 20, None, 1), (7, 5, None, 1), (6, 6, [2] * 6, 2), (5, 3, [1] * 6, 4)):
            res, passes, sched = _stream_run(ctx, trks, nl, scales.copy(), slots, sslots, rounds, None, waves)
            _check(res, ref, len(trks), nl)
            if rounds is not None:
                assert max(r.passes for r in res.values()) > 1  # problems really were carried
        # the tick engine: every resident problem advances one round per tick, admission and retirement on the device.
        # (a) a pool as large as the job; (b) a small pool, few ticks per advance: slots are refilled inside an advance and
        # across advances; (c) one tick per advance, submissions in waves; (d) many ticks: problems admitted AND retired
        # inside one advance
        # (advances are pipelined -- results surface one advance late -- except in the last case, where every advance is read
        # back before the call returns)
        # (ticks = 0: the stream sizes its advances itself from the retired problems' mean life)
        for slots, sslots, ticks, waves, pipelined in ((20, 20, 16, 1, True), (7, 5, 5, 1, True), (6, 4, 1, 3, True), (3, 2, 200, 2, True), (7, 5, 5, 2, False),
                                                      (6, 5, 0, 4, True), (20, 8, 0, 1, False)):
            res, passes, sched = _stream_run(ctx, trks, nl, scales.copy(), slots, sslots, None, None, waves, engine=1, ticks=ticks, pipelined=pipelined)
            _check(res, ref, len(trks), nl)
    finally:
        ctx.set_streams(1)


def test_stream_with_fixed_schedule_and_mixed_coarsest_levels(ctx):
    """the benchmark schedule (1 + K rounds per level) and problems that start on different levels in one pool"""
    from direct_stereo_slam_amd.tracker import Stream, default_params

    p = default_params()
    p.fixed_schedule = 3
    scs = [make_scene("small", seed=760 + i) for i in range(6)]
    nl = scs[0].nl
    trks = [hip_tracker(ctx, sc, p) for sc in scs]
    n = len(trks)
    ref = ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
    for engine in (0, 1):
        st = Stream(ctx, 4, 0, engine)
        tk = st.submit_track(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
        st.drain()
        got = {r.ticket: r for r in st.results()}
        for i, t in enumerate(tk):
            assert np.array_equal(np.array(got[t].pose), ref[1][i])
            assert engine == 1 or got[t].passes == 1  # passes: exactly one pass per problem
        st.close()
    # different starting levels: each against its own single call
    trks = [hip_tracker(ctx, sc) for sc in scs]
    for engine in (0, 1):
        st = Stream(ctx, 3, 0, engine)
        tks = []
        for i, t in enumerate(trks):
            tks += st.submit_track([t], [S.IDENTITY_POSE], np.zeros((1, 2)), nl - 1 - (i % 2))
        st.drain()
        got = {r.ticket: r for r in st.results()}
        for i, t in enumerate(trks):
            good, pose, aff, last = t.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], nl - 1 - (i % 2))
            r = got[tks[i]]
            assert bool(r.good) == bool(good) and np.array_equal(np.array(r.pose), pose)
            assert np.array_equal(np.array(r.last_residuals), np.asarray(last), equal_nan=True)
        st.close()


@pytest.mark.parametrize("engine", [0, 1])
def test_stream_argument_checks(ctx, engine):
    from direct_stereo_slam_amd._lib import DsmError
    from direct_stereo_slam_amd.tracker import Stream

    sc = make_scene("small", seed=790)
    trk = hip_tracker(ctx, sc)
    st = Stream(ctx, 2, 0, engine)
    with pytest.raises(DsmError):
        st.submit_scale([trk], np.ones(1, np.float32), sc.nl - 1)  # no scale slots
    with pytest.raises(DsmError):
        st.submit_track([trk], [S.IDENTITY_POSE], np.zeros((1, 2)), sc.nl)  # coarsest level out of range
    other = make_scene("medium", seed=791)
    st.submit_track([trk], [S.IDENTITY_POSE], np.zeros((1, 2)), sc.nl - 1)
    with pytest.raises(DsmError):
        st.submit_track([hip_tracker(ctx, other)], [S.IDENTITY_POSE], np.zeros((1, 2)), other.nl - 1)  # another geometry
    with pytest.raises(DsmError):
        st.set_engine(engine, -2)
    st.set_engine(engine, 12)  # a fixed number of ticks per advance ...
    st.set_engine(engine, -1)  # ... and back to the stream's own choice
    st.advance()  # an empty pass after everything retired is a no-op
    st.drain()
    assert st.counts()[0] == 0 and len(st.results()) == 1
    st.advance()
    st.close()
