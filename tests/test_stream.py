"""The streaming form of the batched calls (dsm_stream_*, csrc/stream_capi.hip): problems are admitted as slots free up, advance
in passes with a bounded number of rounds per level, are carried over when they need more and retire individually.  None of
that may change a result: every problem's pose, residuals, flags and per-level evaluation counts must equal the batch
calls' bit for bit, whatever the pool size, the rounds per pass or the stream groups."""
import numpy as np
import pytest

from _scenes import S, hip_tracker, make_relief_frames, make_scene, oracle_tracker

pytestmark = pytest.mark.gpu


def _batch_reference(ctx, trks, nl, scales):
    n = len(trks)
    good, poses, affs, last, flow = ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
    ev_t = list(ctx.stats().evals)
    err, sc = ctx.optimize_scale_batch(trks, scales, nl - 1)
    ev_s = list(ctx.stats().evals)
    return good, poses, affs, last, flow, err, sc, ev_t, ev_s


def _stream_run(ctx, trks, nl, scales, track_slots, scale_slots, rounds=None, quantile=None, waves=1, engine=0, ticks=0, pipelined=True, chain=None):  # noqa: PLR0913
    from direct_stereo_slam_amd.tracker import Stream

    n = len(trks)
    st = Stream(ctx, track_slots, scale_slots, engine, ticks)
    chain_seq = None
    if isinstance(chain, (list, tuple)):  # another setting before every advance
        chain_seq = list(chain)
    elif chain is not None:
        st.set_chain(chain)
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
        if chain_seq:
            st.set_chain(chain_seq[passes % len(chain_seq)])
        st.advance()
        passes += 1
        out += st.results()
    while True:
        resident, waiting, _ = st.counts()
        if resident == 0 and waiting == 0:
            break
        assert resident <= track_slots + scale_slots
        if chain_seq:
            st.set_chain(chain_seq[passes % len(chain_seq)])
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


def test_tick_engine_chains_are_scheduling_only(ctx):
    """dsm_stream_set_chain: a problem whose pending evaluation is ONE chunk is evaluated AND stepped by one workgroup inside the
    tick's evaluation launch, for up to max_rounds LM rounds in a row (the reference's loop, TrackerAndScaler.cpp:505-593, running in
    place on the small levels).  Same chunk, same partial, same reduction order: every pose, residual, flag and per-level evaluation
    count must equal the batch calls' -- with chains off, one round, a few, without bound, switched between advances, with slots
    refilled inside a chain's tick, pose and scale problems alike -- and a frame must live fewer ticks."""
    from direct_stereo_slam_amd.tracker import Stream

    # dense templates of the small pyramid: level 0 has several chunks, levels 1-2 one; sparse templates: one chunk on every level
    scs = [make_scene("small", seed=900 + i, template="dense" if i % 2 else "sparse", n0=2500) for i in range(16)]
    nl = scs[0].nl
    trks = [hip_tracker(ctx, sc) for sc in scs]
    n = len(trks)
    scales = np.linspace(0.85, 1.25, n).astype(np.float32)
    ref = _batch_reference(ctx, trks, nl, scales)
    for slots, sslots, ticks, waves, pipelined, chain in ((16, 16, 0, 1, True, 0), (16, 16, 0, 1, True, 1), (16, 16, 0, 1, True, 3), (16, 16, 0, 1, True, 4096),
                                                        (5, 3, 3, 2, True, 4096), (4, 2, 1, 3, False, 2), (6, 4, 2, 2, True, [0, 4096, 1, 0, 5]), (3, 3, 64, 4, True, -1)):
        res, _, _ = _stream_run(ctx, trks, nl, scales.copy(), slots, sslots, None, None, waves, engine=1, ticks=ticks, pipelined=pipelined, chain=chain)
        _check(res, ref, n, nl)
    # a frame's life in ticks: with chains the stream needs fewer ticks for the same problems (one tick per advance: advances = ticks)
    need = {}
    for chain in (0, 4096):
        st = Stream(ctx, n, n, 1, 1)
        st.set_chain(chain)
        st.submit_track(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
        adv = 0
        while True:
            st.advance()
            st.sync()
            adv += 1
            resident, waiting, _ = st.counts()
            if resident == 0 and waiting == 0:
                break
            assert adv < 500
        need[chain] = adv
        st.close()
    assert need[4096] < need[0], need


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


def _oracle_results(frames, kf_every):
    out = []
    for i, sc in enumerate(frames):
        orc = oracle_tracker(sc)
        good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
        ev = list(orc.eval_counts()[0])
        scale = None
        if i % kf_every == 0:
            err, s = orc.optimize_scale(1.0, sc.nl - 1)
            scale = (err, s, list(orc.eval_counts()[0]))
        out.append((bool(good), np.asarray(pose), np.asarray(aff), np.asarray(last), ev, scale))
    return out


def _stream_vs_oracle(ctx, frames, track_slots, scale_slots, kf_every, waves, max_route_flips, engine=1):
    """frames through the tick engine as bench.py drives it (three stream groups, the stream's own tick sizing, pipelined
    advances, a pool smaller than the job, submissions in waves) against orc.track / orc.optimize_scale frame by frame
    (TrackerAndScaler.cpp:451-638, 854-964): flags equal, per-level evaluation counts equal, poses and scales to 1e-4.  A frame
    whose LM route parts from the oracle's (an accept or break test decided by the last bits of a float sum: DESIGN.md 4.4,
    tests/test_sweep_parity.py's 4 % bar) is listed; at most `max_route_flips` of them, and they must still end at the same pose
    to 1e-3."""
    from direct_stereo_slam_amd.tracker import Stream

    nl = frames[0].nl
    ref = _oracle_results(frames, kf_every)
    ctx.set_streams(3)
    try:
        trks = [hip_tracker(ctx, sc) for sc in frames]
        st = Stream(ctx, track_slots, scale_slots, engine, 0)
        owner, got = {}, []
        per = (len(frames) + waves - 1) // waves
        for wv in range(waves):
            idx = list(range(wv * per, min(len(frames), (wv + 1) * per)))
            for i, tk in zip(idx, st.submit_track([trks[i] for i in idx], np.tile(S.IDENTITY_POSE, (len(idx), 1)), np.zeros((len(idx), 2)), nl - 1)):
                owner[tk] = ("track", i)
            kf = [i for i in idx if i % kf_every == 0]
            if kf:
                for i, tk in zip(kf, st.submit_scale([trks[i] for i in kf], np.ones(len(kf), np.float32), nl - 1)):
                    owner[tk] = ("scale", i)
            st.advance()
            got += st.results()
        st.drain()
        got += st.results()
        stats = st.stats()
        st.close()
    finally:
        ctx.set_streams(1)
    res = {}
    for r in got:
        assert r.ticket in owner and owner[r.ticket] not in res  # every problem retires exactly once
        res[owner[r.ticket]] = r
    assert len(res) == len(owner)
    flips = []
    for i, (good, pose, aff, last, ev, scale) in enumerate(ref):
        r = res[("track", i)]
        assert r.kind == 0 and bool(r.good) == good, i
        same = list(r.evals)[:nl] == ev[:nl]
        if not same:
            flips.append((i, list(r.evals)[:nl], ev[:nl]))
        tol = 1e-4 if same else 1e-3
        np.testing.assert_allclose(np.array(r.pose), pose, rtol=0, atol=tol * max(1.0, np.abs(pose[4:]).max()), err_msg=f"frame {i}")
        np.testing.assert_allclose(np.array(r.aff), aff, rtol=10 * tol, atol=10 * tol, err_msg=f"frame {i}")
        if same:
            # last_residuals[l] = sqrt(E / n) of the level's last accepted evaluation; the oracle's E is the reference's SEQUENTIAL float
            # sum (quirk Q1), good to n * 2^-24 relative in the worst case (half of that on the square root): 1e-4 on the small
            # levels, the rounding bound where it is larger (472 720 terms at level 0)
            n_l = np.array([len(frames[i].tpl[0][l]) for l in range(nl)], np.float64)
            assert np.all(np.abs(np.array(r.last_residuals)[:nl] - last[:nl]) <= (1e-4 + 0.25 * n_l * 2.0 ** -24) * last[:nl]), (i, r.last_residuals, last)
        assert np.all(np.isnan(np.array(r.last_residuals)[nl:]))
        # both paths end at the ground truth (relief scenes: the reference algorithm converges from the identity guess)
        np.testing.assert_allclose(np.array(r.pose)[4:], frames[i].gt_pose[4:], atol=5e-3)
        if scale is not None:
            q = res[("scale", i)]
            err_o, s_o, ev_s = scale
            assert q.kind == 1 and abs(q.scale - s_o) < 1e-4 * abs(s_o), i
            # (err: the oracle's E is the reference's sequential float sum, quirk Q1 -- good to n * 2^-24)
            assert abs(q.err - err_o) < max(5e-4, 0.5 * len(frames[i].tpl[0][0]) * 2.0 ** -24) * err_o, i
            if list(q.evals)[:nl] != ev_s[:nl]:
                flips.append((i, "scale", list(q.evals)[:nl], ev_s[:nl]))
    assert len(flips) <= max_route_flips, flips
    return flips, stats


def test_tick_engine_on_the_bench_workload_equals_the_oracle(ctx):
    """the path the bench times -- S2 (1248 x 384, six levels, dense template: 472 720 points, 116 chunks of
    16 points per thread at level 0), RELIEF scenes, dsm_stream_* tick engine with mixed-level item lists, on-device admission,
    result rings, pipelined read-back, three stream groups, a pool smaller than the job -- next to the CPU oracle, frame by
    frame; then two S3 frames (1920 x 1080, six floor-halved levels)."""
    frames = make_relief_frames("kitti6", 16, 3)
    assert len(frames[0].tpl[0][0]) == 1244 * 380 and frames[0].nl == 6
    flips, stats = _stream_vs_oracle(ctx, frames, track_slots=6, scale_slots=2, kf_every=5, waves=2, max_route_flips=1)
    hd = make_relief_frames("hd6", 2, 1, seed0=0x5EED0100)
    _stream_vs_oracle(ctx, hd, track_slots=1, scale_slots=1, kf_every=1, waves=1, max_route_flips=0)


def test_stream_groups_get_hardware_queues_of_their_own(built):
    """The three stream groups and the scale segment of a stream must sit on different hardware queues, or their launches serialise
    (round 5 lost 6-30 % to two groups on one queue before the probe existed: DESIGN.md 4.3a).  A fresh context -- the session's has
    sixteen groups' streams from other tests, more than the runtime has queues --, three groups, a small tick-engine job; the probe's
    outcome as dsm_context_stream_queues reports it, and the results still equal the batch calls."""
    from direct_stereo_slam_amd.tracker import Context

    c = Context(0)
    try:
        assert c.stream_queues() == (1, 0)
        scs = [make_scene("small", seed=760 + i) for i in range(9)]
        trks = [hip_tracker(c, sc) for sc in scs]
        scales = np.linspace(0.9, 1.2, len(trks)).astype(np.float32)
        ref = _batch_reference(c, trks, scs[0].nl, scales)
        c.set_streams(3)
        res, _, _ = _stream_run(c, trks, scs[0].nl, scales.copy(), 9, 4, None, None, 1, engine=1, ticks=0)
        _check(res, ref, len(trks), scs[0].nl)
        assert c.stream_queues() == (4, 0)
    finally:
        c.close()


def test_tick_engine_relief_golden_fixture(ctx):
    """tests/golden/tracker_relief_small.npz (minted from the oracle by make_golden.py: four camera-byte frames of one relief
    texture, 308 x 92 x 3) through the tick engine with two track slots: the stored flags, poses, residuals, evaluation counts"""
    import os

    from direct_stereo_slam_amd.tracker import Stream, TrackerAndScaler

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracker_relief_small.npz"))
    w, h, nl, K, T = int(g["w"]), int(g["h"]), int(g["nl"]), tuple(g["K"]), g["T"]
    tpl = [[g[f"tpl_{name}{l}"] for l in range(nl)] for name in ("u", "v", "id", "c")]
    n = int(g["n_frames"])
    trks = []
    for i in range(n):
        trk = TrackerAndScaler(ctx, w, h, nl, T, K)
        trk.makeK(*K)
        trk.setCoarseTrackingRef(i, (0.0, 0.0), 1.0, *tpl)
        trk.upload_image(0, g["new_u8"][i], 1.0)  # camera bytes in, device makeImages
        trk.upload_image(1, g["right_u8"], 1.0)
        trks.append(trk)
    for engine in (1,):
        st = Stream(ctx, 2, 1, engine, 0)
        tk = st.submit_track(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
        ts = st.submit_scale(trks[:1], np.ones(1, np.float32), nl - 1)
        st.drain()
        got = {r.ticket: r for r in st.results()}
        st.close()
        for i in range(n):
            r = got[tk[i]]
            assert bool(r.good) == bool(g["track_good"][i])
            assert list(r.evals)[:nl] == list(g["track_evals"][i][:nl])
            np.testing.assert_allclose(np.array(r.pose), g["track_pose"][i], rtol=0, atol=1e-4)
            np.testing.assert_allclose(np.array(r.aff), g["track_aff"][i], rtol=1e-3, atol=1e-3)
            np.testing.assert_allclose(np.array(r.last_residuals)[:nl], g["track_last"][i][:nl], rtol=1e-4)
        q = got[ts[0]]
        assert abs(q.scale - float(g["scale_out"])) < 1e-4 and abs(q.err - float(g["scale_err"])) < 5e-4 * float(g["scale_err"])
        assert list(q.evals)[:nl] == list(g["scale_evals"][:nl])


@pytest.mark.parametrize("engine", [0, 1])
def test_stream_engines_with_an_empty_level_and_a_single_point_level(ctx, engine):
    """a level without template points still takes its LM step (TrackerAndScaler.cpp:474-505 on n = 0: the level ends at once, its
    residual is 0 / 0): both engines must return the batch call's bits, and those the oracle's NaN at the empty level.  (Round 6: the fused
    evaluate-and-step launch of the batch form reduced over a stale point count there -- the speculative row's -- until this test.)"""
    from direct_stereo_slam_amd.tracker import Stream

    scs = [make_scene("small", seed=820 + i) for i in range(6)]
    for i, sc in enumerate(scs):
        for a in sc.tpl:
            a[1] = a[1][:0 if i % 2 == 0 else 1].copy()  # level 1: empty / one point
            if i == 3:
                a[2] = a[2][:0].copy()  # the coarsest (starting) level empty as well
    nl = scs[0].nl
    trks = [hip_tracker(ctx, sc) for sc in scs]
    n = len(trks)
    scales = np.ones(n, np.float32)
    ref = _batch_reference(ctx, trks, nl, scales)
    res, _, _ = _stream_run(ctx, trks, nl, scales.copy(), 4, 3, None, None, 2, engine=engine, ticks=0)
    _check(res, ref, n, nl)
    for i, sc in enumerate(scs):  # and the oracle, the single fused-launch call included
        good_o, pose_o, aff_o, last_o, _ = oracle_tracker(sc).track(S.IDENTITY_POSE, [0, 0], nl - 1)
        good_g, pose_g, aff_g, last_g = trks[i].trackNewestCoarse(S.IDENTITY_POSE, [0, 0], nl - 1)
        for last in (ref[3][i], np.asarray(last_g)):
            assert np.array_equal(np.isnan(last[:nl]), np.isnan(last_o[:nl])), (i, last, last_o)
            np.testing.assert_allclose(last[:nl], last_o[:nl], rtol=1e-3, equal_nan=True)
        assert bool(ref[0][i]) == good_o == bool(good_g)
        np.testing.assert_allclose(ref[1][i], pose_o, atol=2e-4)
