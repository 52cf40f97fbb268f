"""Pins the C oracle against a SECOND, independently written restatement (oracle/numpy_ref.py: vectorised numpy float32
written from the reference's text, matrix-exponential SE3, its own pivoted LDLT): per evaluation the Vec6 rs, H, b agree
BIT FOR BIT (both follow TrackerAndScaler.cpp:640-852,966-1172 with IEEE float32 and no contraction); per LM run the
evaluation and acceptance counts per level, the tracked flag and the results agree (poses to 1e-9: the two SE3 exponentials
differ in the last bits of a double).  Removes single-author risk; does not replace a reference binary (parity stays
"unpinned")."""
import numpy as np
import pytest

from _scenes import make_scene, oracle_tracker
from direct_stereo_slam_amd import synth as S
from oracle import numpy_ref as N


def numpy_tracker(sc, **kw):
    t = N.NumpyTracker(sc.w, sc.h, sc.nl, sc.T, sc.K, **kw)
    t.make_k(*sc.K)
    t.set_ref(0.0, 0.0, 1.0, *sc.tpl)
    t.set_frame(0, sc.new_p, 1.0)
    t.set_frame(1, sc.right_p, 1.0)
    return t


@pytest.mark.parametrize("size,template", [("tiny", "dense"), ("small", "dense"), ("small", "sparse"), ("medium", "dense")])
def test_single_evaluations_bit_identical(built, size, template):
    sc = make_scene(size, seed=11, template=template, n0=3000)
    orc, npt = oracle_tracker(sc), numpy_tracker(sc)
    for lvl in range(sc.nl):
        for pose, aff in [(S.IDENTITY_POSE, [0.0, 0.0]), (sc.gt_pose, list(sc.gt_aff))]:
            for cutoff in (20.0, 5.0):
                rs_o = orc.calc_res_pose(lvl, pose, aff, cutoff)
                H_o, b_o = orc.calc_gs_pose(lvl, pose, aff)
                rs_n = npt.calc_res_pose(lvl, N.pose_to_matrix(np.asarray(pose, float)), aff, cutoff)
                H_n, b_n, n_n = npt.calc_gs_pose(lvl, aff)
                assert n_n == orc.pose_warped_n()
                np.testing.assert_array_equal(rs_n, rs_o)  # E (sequential float sum), counts, flow indicators, saturation ratio
                np.testing.assert_array_equal(H_n, H_o)    # 4 lanes, 1k / 1m shift-up, lanes added last
                np.testing.assert_array_equal(b_n, b_o)
        for scale in (1.0, 0.8, 5.0):
            rs_o = orc.calc_res_scale(lvl, scale, 20.0)
            Hs_o, bs_o = orc.calc_gs_scale(lvl, scale)
            rs_n = npt.calc_res_scale(lvl, scale, 20.0)
            Hs_n, bs_n, n_n = npt.calc_gs_scale(lvl, scale)
            assert n_n == orc.scale_warped_n()
            np.testing.assert_array_equal(rs_n, rs_o)
            assert (np.float32(Hs_n), np.float32(bs_n)) == (np.float32(Hs_o), np.float32(bs_o))


@pytest.mark.parametrize("size,seed,motion", [("tiny", 3, 1.0), ("small", 5, 1.0), ("small", 7, 3.0), ("medium", 9, 1.0)])
def test_lm_drivers_agree(built, size, seed, motion):
    sc = make_scene(size, seed=seed, motion_scale=motion)
    orc, npt = oracle_tracker(sc), numpy_tracker(sc)
    good_o, pose_o, aff_o, last_o, flow_o = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    res_o, gs_o = orc.eval_counts()
    good_n, pose_n, aff_n, last_n, flow_n = npt.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    assert good_n == good_o
    assert npt.res_evals == res_o[:sc.nl]  # same number of residual evaluations on every level ...
    accepts = [sum(1 for l, a in npt.trace if l == lvl and a) for lvl in range(sc.nl)]
    init_evals = [res_o[l] - sum(1 for ll, _ in npt.trace if ll == l) for l in range(sc.nl)]  # cut-off repeats + level repeats
    assert [a + i for a, i in zip(accepts, init_evals)] == gs_o[:sc.nl]  # ... and the same number of accepted steps
    np.testing.assert_allclose(pose_n, pose_o, atol=1e-5)
    np.testing.assert_allclose(aff_n, aff_o, atol=2e-4)
    np.testing.assert_allclose(last_n[:sc.nl], last_o[:sc.nl], rtol=1e-5)
    np.testing.assert_allclose(flow_n, flow_o, rtol=1e-4)
    err_o, s_o = orc.optimize_scale(1.1, sc.nl - 1)
    res_so, gs_so = orc.eval_counts()
    err_n, s_n = npt.optimize_scale(1.1, sc.nl - 1)
    assert npt.res_evals == res_so[:sc.nl]
    assert np.float32(s_n) == np.float32(s_o) and np.float32(err_n) == np.float32(err_o)  # the float 1-DoF problem: bit for bit


def test_fixed_affine_modes_and_abort(built):
    """the 6- and 7-dimensional sub-solves (TrackerAndScaler.cpp:511-534) and the abort test (:598)"""
    sc = make_scene("small", seed=5)
    for ma, mb in ((-1.0, -1.0), (0.0, -1.0), (-1.0, 0.0)):
        p = __import__("oracle.oracle", fromlist=["x"]).default_params()
        p.affine_opt_mode_a, p.affine_opt_mode_b = ma, mb
        orc, npt = oracle_tracker(sc, p), numpy_tracker(sc, mode_a=ma, mode_b=mb)
        good_o, pose_o, aff_o, last_o, _ = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
        good_n, pose_n, aff_n, last_n, _ = npt.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
        assert good_n == good_o and npt.res_evals == orc.eval_counts()[0][:sc.nl]
        np.testing.assert_allclose(pose_n, pose_o, atol=1e-5)
        np.testing.assert_allclose(aff_n, aff_o, atol=2e-4)
    orc, npt = oracle_tracker(sc), numpy_tracker(sc)
    mr = np.full(6, 1e-3)  # every level's residual is far above 1.5 * 1e-3: abort at the coarsest level
    good_o = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1, mr)[0]
    good_n = npt.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1, mr)[0]
    assert good_o is False and good_n is False and npt.res_evals == orc.eval_counts()[0][:sc.nl]
