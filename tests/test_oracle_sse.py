"""The SSE-intrinsics form of calcGSSSEPose / calcGSSSEScale (oracle/dsm_oracle_sse.c: the reference's own loop
structure, TrackerAndScaler.cpp:640-697,966-1005, ScaleAccumulator.h:60-105 -- the TIMED CPU baseline of bench.py)
against the scalar lane emulation of the parity oracle: bit-identical in the parity build, including the 1k / 1m
shift-up of long buffers (> 1000 packs) and whole track / optimize_scale runs."""
import numpy as np
import pytest

from _scenes import make_scene
from direct_stereo_slam_amd import synth as S
from oracle import oracle as O


def _pair(sc):
    out = []
    for sse in (False, True):
        o = O.OracleTracker(sc.w, sc.h, sc.nl, sc.T, sc.K)
        o.use_sse(sse)
        o.make_k(*sc.K)
        o.set_ref(0, 0.0, 0.0, 1.0, *sc.tpl)
        o.set_frame(0, sc.new_p, 1.0)
        o.set_frame(1, sc.right_p, 1.0)
        out.append(o)
    return out


@pytest.mark.parametrize("size", ["tiny", "small", "medium"])  # medium level 0: 108 k points = 27 shift-ups
def test_gs_pose_and_scale_bit_identical(size):
    sc = make_scene(size, seed=11)
    a, b = _pair(sc)
    for lvl in range(sc.nl):
        for o in (a, b):
            o.calc_res_pose(lvl, S.IDENTITY_POSE, [0.01, 1.0], 20.0)
        Ha, ba = a.calc_gs_pose(lvl, S.IDENTITY_POSE, [0.01, 1.0])
        Hb, bb = b.calc_gs_pose(lvl, S.IDENTITY_POSE, [0.01, 1.0])
        assert a.pose_warped_n() == b.pose_warped_n() > 0
        assert np.array_equal(Ha, Hb) and np.array_equal(ba, bb)
        for o in (a, b):
            o.calc_res_scale(lvl, 1.05, 20.0)
        assert a.calc_gs_scale(lvl, 1.05) == b.calc_gs_scale(lvl, 1.05)


def test_whole_lm_runs_identical():
    sc = make_scene("small", seed=5)
    a, b = _pair(sc)
    ra = a.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    rb = b.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])
    assert a.eval_counts() == b.eval_counts()
    assert a.optimize_scale(1.1, sc.nl - 1) == b.optimize_scale(1.1, sc.nl - 1)
