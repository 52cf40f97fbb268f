"""Row N4 (GPU): the hypothesis loop of FrontEnd::trackNewCoarse (FrontEnd.cpp:132-256), batched on
the device, against the reference's sequential loop run on the CPU oracle."""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S
from oracle import oracle as O

from _scenes import hip_tracker, make_scene, oracle_tracker, regrad
from test_replay_sequence import mul_pose

pytestmark = pytest.mark.gpu


def reference_tries(const_motion):
    """FrontEnd.cpp:150-186: constant / double / half / zero motion, identity, then 26 rotation signs for
    rot_delta = 0.02, 0.03, ... < 0.05 (a float loop)"""
    tries = [const_motion.copy(), mul_pose(const_motion, const_motion), S.IDENTITY_POSE.copy(), S.IDENTITY_POSE.copy(),
             S.IDENTITY_POSE.copy()]
    signs = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (-1, 1, 0),
             (0, -1, 1), (-1, 0, 1), (1, -1, 0), (0, 1, -1), (1, 0, -1), (-1, -1, 0), (0, -1, -1), (-1, 0, -1), (-1, -1, -1),
             (-1, -1, 1), (-1, 1, -1), (-1, 1, 1), (1, -1, -1), (1, -1, 1), (1, 1, -1), (1, 1, 1)]
    rd = np.float32(0.02)
    while rd < np.float32(0.05):
        for s in signs:
            q = np.array([s[0] * rd, s[1] * rd, s[2] * rd, 1.0], np.float64)
            q /= np.linalg.norm(q)
            tries.append(mul_pose(const_motion, np.concatenate([q, np.zeros(3)])))
        rd = np.float32(rd + np.float32(0.01))
    return np.array(tries)


def sequential_reference(orc, tries, aff_last, coarsest, last_rmse0, thr=1.5):
    """the loop of FrontEnd.cpp:194-256 verbatim on the oracle tracker"""
    achieved = np.full(6, np.nan)
    have, flow, best_pose, best_aff, used = False, np.array([100.0] * 3), S.IDENTITY_POSE.copy(), np.zeros(2), 0
    for i in range(len(tries)):
        good, pose, aff, cur, fl = orc.track(tries[i], aff_last, coarsest, achieved)
        used += 1
        if good and np.isfinite(np.float32(cur[0])) and not (cur[0] >= achieved[0]):
            flow, best_aff, best_pose, have = fl.copy(), aff.copy(), pose.copy(), True
        if have:
            for l in range(5):
                if not np.isfinite(np.float32(achieved[l])) or achieved[l] > cur[l]:
                    achieved[l] = cur[l]
        if have and achieved[0] < last_rmse0 * thr:
            break
    if not have:
        return False, tries[0], np.asarray(aff_last, float), np.zeros(3), achieved, used
    return True, best_pose, best_aff, flow, achieved, used


@pytest.mark.parametrize("case", ["first_try_wins", "needs_retries", "all_fail"])
def test_batched_hypotheses_match_sequential_reference(ctx, case):
    from direct_stereo_slam_amd.tracker import track_hypotheses

    if case == "first_try_wins":
        sc = make_scene("small", seed=81)
        const_motion, last_rmse0 = sc.gt_pose.copy(), 100.0
    elif case == "needs_retries":
        # large rotation: the constant-motion guess is far off, a rotated hypothesis has to win; a small
        # last_coarse_rmse keeps the loop going through many tries
        sc = make_scene("small", seed=82, motion_scale=4.0)
        const_motion, last_rmse0 = S.IDENTITY_POSE.copy(), 0.5
    else:
        sc = make_scene("small", seed=83)
        sc.new_p = [regrad(np.full_like(p, np.nan)) for p in sc.new_p]  # a frame with no usable texel: every try fails
        const_motion, last_rmse0 = S.IDENTITY_POSE.copy(), 1.0
    tries = reference_tries(const_motion)
    assert len(tries) in (83, 109)
    orc, trk = oracle_tracker(sc), hip_tracker(ctx, sc)
    ref = sequential_reference(orc, tries, [0.0, 0.0], sc.nl - 1, last_rmse0)
    got = track_hypotheses(ctx, trk, tries, [0.0, 0.0], sc.nl - 1, last_rmse0)
    assert got[0] == ref[0]
    np.testing.assert_allclose(got[1], ref[1], atol=1e-4)
    np.testing.assert_allclose(got[2], ref[2], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(got[3], ref[3], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(got[4][:sc.nl], ref[4][:sc.nl], rtol=1e-4, equal_nan=True)
    if case == "first_try_wins":
        assert got[5] == ref[5] == 1
    if case == "needs_retries":
        assert ref[5] > 5  # the sequential loop really went through several hypotheses
    if case == "all_fail":
        assert not got[0] and ref[5] == len(tries)
