"""Row N2: PoseEstimator (loop-closure direct alignment, PoseEstimator.cpp:84-506).
CPU: the oracle restatement recovers a known relative pose.  GPU: the HIP path (same eval/LM
kernels in mode 2) against the oracle."""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S
from oracle import oracle as O

from _scenes import make_scene


def bilinear(img, x, y):
    ix, iy = np.floor(x).astype(int), np.floor(y).astype(int)
    dx, dy = (x - ix).astype(np.float32), (y - iy).astype(np.float32)
    return (dx * dy * img[iy + 1, ix + 1] + (dy - dx * dy) * img[iy + 1, ix] + (dx - dx * dy) * img[iy, ix + 1]
            + (1 - dx - dy + dx * dy) * img[iy, ix]).astype(np.float32)


def loop_inputs(sc, n=1500, seed=0):
    """what LoopHandler::publishKeyframes stores per keyframe (LoopHandler.cpp:166-181): 3-D points in the
    keyframe and their reference intensity on every pyramid level"""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = sc.K
    u = rng.uniform(4, sc.w - 5, n)
    v = rng.uniform(4, sc.h - 5, n)
    idl0 = sc.scene.idepth(sc.K, sc.w, sc.h)
    idp = bilinear(idl0, u, v).astype(np.float64)
    xyz = np.stack([(u - cx) / fx / idp, (v - cy) / fy / idp, 1 / idp], 1)
    cols = []
    for l in range(sc.nl):
        ul, vl = (u + 0.5) / (1 << l) - 0.5, (v + 0.5) / (1 << l) - 0.5
        cols.append(bilinear(sc.ref_p[l][..., 0], np.clip(ul, 0, (sc.w >> l) - 2), np.clip(vl, 0, (sc.h >> l) - 2)))
    return xyz, cols


def gt_matrix(sc):
    T = np.eye(4)
    T[:3, :3] = S.quat_to_rot(sc.gt_pose[:4])
    T[:3, 3] = sc.gt_pose[4:]
    return T


def test_oracle_pose_estimator_recovers_ground_truth():
    sc = make_scene("small", seed=71, a=0.0, b=0.0)
    xyz, cols = loop_inputs(sc)
    pe = O.OraclePoseEstimator(sc.w, sc.h, sc.nl)
    ok, T, err, inl = pe.estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, np.eye(4))
    assert ok and inl > 90 and err < 5.0
    np.testing.assert_allclose(T, gt_matrix(sc), atol=6e-3)
    # a hopeless initial guess is rejected by the acceptance test (:484-505)
    bad = np.eye(4)
    bad[:3, 3] = [3.0, 0.0, 0.0]
    ok2, _, err2, inl2 = pe.estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, bad)
    assert not ok2


@pytest.mark.gpu
@pytest.mark.parametrize("size,n,seed", [("small", 1500, 72), ("medium", 4000, 73), ("small", 40, 74)])
def test_hip_pose_estimator_matches_oracle(ctx, size, n, seed):
    from direct_stereo_slam_amd.tracker import PoseEstimator

    sc = make_scene(size, seed=seed, a=0.01, b=2.0)
    xyz, cols = loop_inputs(sc, n=n, seed=seed)
    orc = O.OraclePoseEstimator(sc.w, sc.h, sc.nl)
    hip = PoseEstimator(ctx, sc.w, sc.h, sc.nl)
    for guess in (np.eye(4), gt_matrix(sc)):
        ok_o, T_o, err_o, inl_o = orc.estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, guess)
        ok_g, T_g, err_g = hip.estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, guess)
        assert ok_g == ok_o
        np.testing.assert_allclose(T_g, T_o, atol=1e-4)
        assert abs(err_g - err_o) <= 1e-4 * err_o
    assert ok_o
    if n >= 1000:  # 40 points constrain the pose only to a few centimetres
        np.testing.assert_allclose(T_g, gt_matrix(sc), atol=1e-2)


@pytest.mark.gpu
def test_hip_pose_estimator_rejects_like_the_oracle(ctx):
    from direct_stereo_slam_amd.tracker import PoseEstimator

    sc = make_scene("small", seed=75)
    xyz, cols = loop_inputs(sc)
    bad = np.eye(4)
    bad[:3, 3] = [3.0, 0.0, 0.0]
    ok_o, T_o, err_o, _ = O.OraclePoseEstimator(sc.w, sc.h, sc.nl).estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, bad)
    ok_g, T_g, err_g = PoseEstimator(ctx, sc.w, sc.h, sc.nl).estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, bad)
    assert not ok_o and not ok_g
    # more points than coarse-level pixels is legal (same point set on every level)
    sc2 = make_scene("tiny", seed=76)
    xyz2, cols2 = loop_inputs(sc2, n=4000)
    assert 4000 > (sc2.w >> 1) * (sc2.h >> 1)
    ok_o, T_o, err_o, _ = O.OraclePoseEstimator(sc2.w, sc2.h, sc2.nl).estimate(xyz2, cols2, 1.0, sc2.new_p, 1.0, sc2.K, sc2.nl - 1, np.eye(4))
    ok_g, T_g, err_g = PoseEstimator(ctx, sc2.w, sc2.h, sc2.nl).estimate(xyz2, cols2, 1.0, sc2.new_p, 1.0, sc2.K, sc2.nl - 1, np.eye(4))
    assert ok_g == ok_o
    np.testing.assert_allclose(T_g, T_o, atol=1e-4)
