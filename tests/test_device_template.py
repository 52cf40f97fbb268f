"""Row N3 (A4 on the device): dsm_tracker_set_ref_from_points == oracle makeCoarseDepthL0 -> set_ref, bit for bit
(TrackerAndScaler.cpp:143-327).  Same float32 operations in the same order; the emit order is the
reference's row-major order, so the templates must be identical arrays, not just equal sets."""
import numpy as np
import pytest

from _scenes import O, S, hip_tracker, make_scene, regrad

pytestmark = pytest.mark.gpu


def _points(sc, seed, npts, collisions=True):
    rng = np.random.default_rng(seed)
    pu = rng.uniform(3, sc.w - 4, npts).astype(np.float32)
    pv = rng.uniform(3, sc.h - 4, npts).astype(np.float32)
    idl0 = sc.scene.idepth(sc.K, sc.w, sc.h)
    pid = idl0[(pv + 0.5).astype(int), (pu + 0.5).astype(int)] * rng.uniform(0.9, 1.1, npts).astype(np.float32)
    pw = np.sqrt(1e-3 / (rng.uniform(1e-3, 10, npts) + 1e-12)).astype(np.float32)
    if collisions:
        pid[:5] = -0.1  # non-positive idepths are dropped at the emit stage (:302)
        for a, b in ((10, 14), (100, 109), (500, 502)):  # several points on one pixel: summed in point order (:160-161)
            if b < npts:
                pu[a:b] = pu[a]
                pv[a:b] = pv[a]
        # the same pixel hit by points far apart in the list
        pu[npts - 1], pv[npts - 1] = pu[10], pv[10]
    return pu, pv, pid, pw


@pytest.mark.parametrize("size,npts", [("small", 1500), ("medium", 6000), ("kitti", 14000), ("tiny", 60)])
def test_set_ref_from_points_matches_oracle(ctx, size, npts):
    sc = make_scene(size, seed=81)
    pu, pv, pid, pw = _points(sc, 81, npts)
    ref = [p.copy() for p in sc.ref_p]
    ref[0][min(40, sc.h - 5), min(50, sc.w - 5), 0] = np.nan  # non-finite reference colour is dropped (:302)
    ref[0] = regrad(ref[0])
    orc = O.OracleTracker(sc.w, sc.h, sc.nl, sc.T, sc.K)
    orc.make_k(*sc.K)
    exp = orc.make_coarse_depth_l0(pu, pv, pid, pw, ref)
    trk = hip_tracker(ctx, sc)
    trk.upload_frame(0, ref, 1.0)  # the keyframe's pyramid lives in slot 0 of the tracker that tracked it
    n = trk.setCoarseTrackingRefFromPoints(7, (0.01, 2.0), 1.25, pu, pv, pid, pw)
    assert trk.refFrameID == 7
    assert n == [len(a) for a in exp[0]]
    assert n[0] > 0
    for l in range(sc.nl):
        got = trk.get_template(l)
        for k in range(4):
            np.testing.assert_array_equal(got[k], exp[k][l])


def test_template_built_on_device_tracks_like_the_uploaded_one(ctx):
    """the two ways of installing a reference (host lists vs device construction) give bit-identical tracking"""
    from direct_stereo_slam_amd.tracker import make_coarse_depth_l0

    sc = make_scene("medium", seed=82)
    pu, pv, pid, pw = _points(sc, 82, 8000, collisions=False)
    tpl = make_coarse_depth_l0(sc.w, sc.h, sc.nl, pu, pv, pid, pw, sc.ref_p)
    a, b = hip_tracker(ctx, sc), hip_tracker(ctx, sc)
    a.setCoarseTrackingRef(3, (0.0, 0.0), 1.0, *tpl)
    a.upload_frame(0, sc.new_p, 1.0)
    # b: the keyframe pyramid sits in another tracker's slot (the reference swaps two trackers, FrontEnd.cpp:627-632)
    owner = hip_tracker(ctx, sc)
    owner.upload_frame(0, sc.ref_p, 1.0)
    b.setCoarseTrackingRefFromPoints(3, (0.0, 0.0), 1.0, pu, pv, pid, pw, frame_owner=owner, slot=0)
    b.upload_frame(0, sc.new_p, 1.0)
    ra = a.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    rb = b.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    assert ra[0] and rb[0]
    for x, y in zip(ra[1:], rb[1:]):
        np.testing.assert_array_equal(x, y)


def test_set_ref_from_points_errors(ctx):
    from direct_stereo_slam_amd._lib import DsmError

    sc = make_scene("tiny", seed=83)
    trk = hip_tracker(ctx, sc)
    pu, pv, pid, pw = _points(sc, 83, 20, collisions=False)
    fresh = type(trk)(ctx, sc.w, sc.h, sc.nl, sc.T, sc.K)
    fresh.makeK(*sc.K)
    with pytest.raises(DsmError):  # no pyramid in the owner's slot yet
        trk.setCoarseTrackingRefFromPoints(1, (0, 0), 1.0, pu, pv, pid, pw, frame_owner=fresh)
    pu[3] = sc.w + 5.0  # the reference would write out of bounds (:160): rejected, reference invalidated
    with pytest.raises(DsmError):
        trk.setCoarseTrackingRefFromPoints(1, (0, 0), 1.0, pu, pv, pid, pw)
    with pytest.raises(DsmError):
        trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], 1)
    # empty window: every level empty, still a valid (if useless) reference
    n = trk.setCoarseTrackingRefFromPoints(1, (0, 0), 1.0, pu[:0], pv[:0], pid[:0], pw[:0])
    assert n == [0] * sc.nl


def test_batched_set_refs_equal_the_single_calls(ctx):
    """dsm_set_refs_from_points: the new keyframes of several trackers in one call (one host synchronisation) -- every template
    equal to the one the single call builds, bit for bit; different point counts per job; a tracker twice in one call is refused"""
    from direct_stereo_slam_amd._lib import DsmError

    sc = make_scene("small", seed=84)
    jobs, single = [], []
    for j, npts in enumerate((1500, 40, 2600, 0)):
        pu, pv, pid, pw = _points(sc, 840 + j, max(npts, 1), collisions=npts > 600)
        pu, pv, pid, pw = pu[:npts], pv[:npts], pid[:npts], pw[:npts]
        a, b = hip_tracker(ctx, sc), hip_tracker(ctx, sc)
        for t in (a, b):
            t.upload_frame(0, sc.ref_p, 1.0)
        single.append((a, a.setCoarseTrackingRefFromPoints(10 + j, (0.01 * j, 1.0), 1.0 + j, pu, pv, pid, pw)))
        jobs.append({"tracker": b, "ref_frame_id": 10 + j, "ref_aff": (0.01 * j, 1.0), "ref_exposure": 1.0 + j, "pu": pu, "pv": pv, "pidepth": pid, "pweight": pw})
    ns = type(jobs[0]["tracker"]).setCoarseTrackingRefsFromPoints(ctx, jobs)
    for (a, n_a), job, n_b in zip(single, jobs, ns):
        b = job["tracker"]
        assert n_a == n_b and b.refFrameID == a.refFrameID
        for l in range(sc.nl):
            for x, y in zip(a.get_template(l), b.get_template(l)):
                np.testing.assert_array_equal(x, y)
    with pytest.raises(DsmError):
        type(jobs[0]["tracker"]).setCoarseTrackingRefsFromPoints(ctx, [jobs[0], jobs[0]])
