"""dsm_params.tile_l0: level 0 of a dense template in tile order, the warped tile's window of the target plane staged in LDS.
Per point nothing changes (the window holds the values the gathers fetch), so the integer outputs are those of the row-major
form bit for bit; the float sums are formed in another order (another thread owns a point) and agree to float tolerance; and
every scheduling form -- launches, fused, work queue, the stream's tick and pass engines -- still equals every other bit for
bit under the setting.  Windows that do not fit (large motion, rotated tiles) fall back to the gathers, tile by tile."""
import numpy as np
import pytest

from _scenes import S, hip_tracker, make_scene

pytestmark = pytest.mark.gpu


def _params(tile, **kw):
    from direct_stereo_slam_amd.tracker import default_params

    p = default_params()
    p.tile_l0 = tile
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@pytest.mark.parametrize("size,seed", [("medium", 41), ("kitti6", 42)])
def test_tile_form_evaluation_equals_the_row_major_form(ctx, size, seed):
    sc = make_scene(size, seed=seed)
    a, b = hip_tracker(ctx, sc, _params(0)), hip_tracker(ctx, sc, _params(1))
    rng = np.random.default_rng(seed)
    for k in range(4):  # the identity, the truth, and poses off the truth (windows that move and grow)
        pose = [S.IDENTITY_POSE, sc.gt_pose, None, None][k]
        if pose is None:
            R, t = S.random_motion(rng, sigma_t=np.array((0.05, 0.05, 0.2)) * (1 + 3 * (k - 2)), sigma_r=0.005 * (1 + 8 * (k - 2)))
            pose = S.pose_from_Rt(R, t)
        ra, Ha, ba, na = a.calcResPose(0, pose, [0.01, 1.0], 20.0)
        rb, Hb, bb, nb = b.calcResPose(0, pose, [0.01, 1.0], 20.0)
        assert na == nb and int(ra[1]) == int(rb[1]) and ra[5] == rb[5]  # warped count, numTermsInE, saturated ratio: exact
        np.testing.assert_allclose(rb[0], ra[0], rtol=2e-6)
        np.testing.assert_allclose(rb[[2, 4]], ra[[2, 4]], rtol=1e-6)  # flow indicators: still every 32nd ORIGINAL index
        np.testing.assert_allclose(Hb, Ha, rtol=0, atol=2e-5 * np.abs(Ha).max())
        np.testing.assert_allclose(bb, ba, rtol=0, atol=2e-5 * np.abs(ba).max())
    # the lower levels and the scale evaluation do not change at all
    for lvl in range(1, sc.nl):
        ra, Ha, ba, na = a.calcResPose(lvl, sc.gt_pose, [0, 0], 20.0)
        rb, Hb, bb, nb = b.calcResPose(lvl, sc.gt_pose, [0, 0], 20.0)
        assert np.array_equal(ra, rb) and np.array_equal(Ha, Hb) and np.array_equal(ba, bb)
    assert all(np.array_equal(x, y) for x, y in zip(a.calcResScale(0, 1.0, 20.0)[:3], b.calcResScale(0, 1.0, 20.0)[:3]))


def test_tile_form_tracks_like_the_row_major_form_and_every_schedule_agrees(ctx):
    from direct_stereo_slam_amd.tracker import Stream

    scs = [make_scene("medium", seed=50 + i) for i in range(6)]
    nl = scs[0].nl
    n = len(scs)
    ref_row = ctx.track_batch([hip_tracker(ctx, sc, _params(0, work_queue=0)) for sc in scs], np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
    outs = []
    for kw in (dict(work_queue=0, fuse_lm=0), dict(work_queue=0, fuse_lm=2), dict(work_queue=2), dict(work_queue=0, speculate=2)):
        trks = [hip_tracker(ctx, sc, _params(1, **kw)) for sc in scs]
        outs.append(ctx.track_batch(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1))
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            assert np.array_equal(x, y, equal_nan=True)
    # the row-major form: the same decisions (evaluation counts are part of the stream results below), poses to tolerance
    assert np.array_equal(outs[0][0], ref_row[0])
    np.testing.assert_allclose(outs[0][1], ref_row[1], atol=1e-4)
    trks = [hip_tracker(ctx, sc, _params(1)) for sc in scs]
    for engine in (0, 1):
        st = Stream(ctx, 4, 0, engine)
        tk = st.submit_track(trks, np.tile(S.IDENTITY_POSE, (n, 1)), np.zeros((n, 2)), nl - 1)
        st.drain()
        got = {r.ticket: r for r in st.results()}
        for i, t in enumerate(tk):
            assert np.array_equal(np.array(got[t].pose), outs[0][1][i]) and bool(got[t].good) == bool(outs[0][0][i])
        st.close()
    # scaleCoarseDepthL0 reaches the tile copy too
    t0, t1 = hip_tracker(ctx, scs[0], _params(0)), hip_tracker(ctx, scs[0], _params(1))
    for t in (t0, t1):
        t.scaleCoarseDepthL0(1.3)
    r0, r1 = t0.calcResPose(0, scs[0].gt_pose, [0, 0], 20.0), t1.calcResPose(0, scs[0].gt_pose, [0, 0], 20.0)
    assert r0[3] == r1[3] and int(r0[0][1]) == int(r1[0][1])
    np.testing.assert_allclose(r1[0][0], r0[0][0], rtol=2e-6)
