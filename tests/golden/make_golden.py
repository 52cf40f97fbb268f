#!/usr/bin/env python3
"""Mints the golden fixtures under tests/golden/ from the CPU oracle (oracle/dsm_oracle.c, parity
build).  The reference itself has no tests, golden vectors or buildable sources in this image
(SURVEY.md sections 4 and 8c), so these vectors pin the ORACLE (against accidental change) and the
HIP path (against the oracle) -- they are not outputs of the reference binary.

    python tests/golden/make_golden.py        # rewrites tracker_tiny.npz and ringkey_500.npz

Inputs are seeded synthetic scenes (direct_stereo_slam_amd/synth.py); every array needed to replay
the case is stored next to the expected outputs, so the tests do not depend on numpy's RNG stream.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from direct_stereo_slam_amd import synth as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

from _scenes import make_scene, oracle_tracker  # noqa: E402
from test_oracle_ringkey import ring_keys  # noqa: E402


def tracker_fixture():
    sc = make_scene("tiny", seed=2024)  # 154x46, 2 levels (SURVEY.md section 8c suggestion)
    orc = oracle_tracker(sc)
    out = {"w": sc.w, "h": sc.h, "nl": sc.nl, "K": np.asarray(sc.K, np.float64), "T": sc.T, "gt_pose": sc.gt_pose}
    for l in range(sc.nl):
        for name, arr in zip(("u", "v", "id", "c"), sc.tpl):
            out[f"tpl_{name}{l}"] = arr[l]
        out[f"new{l}"] = sc.new_p[l]
        out[f"right{l}"] = sc.right_p[l]
    evals = []
    for lvl in range(sc.nl):
        for tag, pose, aff in (("id", S.IDENTITY_POSE, np.zeros(2)), ("gt", sc.gt_pose, sc.gt_aff)):
            rs = orc.calc_res_pose(lvl, pose, aff, 20.0)
            e64 = orc.last_energy_f64()
            H, b = orc.calc_gs_pose(lvl, pose, aff)
            out[f"pose_{tag}{lvl}_in"] = np.concatenate([pose, aff])
            out[f"pose_{tag}{lvl}_rs"] = rs
            out[f"pose_{tag}{lvl}_E64"] = e64
            out[f"pose_{tag}{lvl}_H"] = H
            out[f"pose_{tag}{lvl}_b"] = b
            out[f"pose_{tag}{lvl}_n"] = orc.pose_warped_n()
        for s in (1.0, 0.8):
            rs = orc.calc_res_scale(lvl, s, 20.0)
            e64 = orc.last_energy_f64()
            Hs, bs = orc.calc_gs_scale(lvl, s)
            out[f"scale_{s}_{lvl}_rs"] = rs
            out[f"scale_{s}_{lvl}_E64"] = e64
            out[f"scale_{s}_{lvl}_Hb"] = np.array([Hs, bs], np.float32)
            out[f"scale_{s}_{lvl}_n"] = orc.scale_warped_n()
    good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    out["track_good"], out["track_pose"], out["track_aff"], out["track_last"], out["track_flow"] = good, pose, aff, last, flow
    out["track_evals"] = np.array(orc.eval_counts()[0])
    err, s = orc.optimize_scale(1.0, sc.nl - 1)
    out["scale_err"], out["scale_out"] = np.float32(err), np.float32(s)
    out["scale_evals"] = np.array(orc.eval_counts()[0])
    np.savez_compressed(os.path.join(HERE, "tracker_tiny.npz"), **out)


def ringkey_fixture():
    keys = ring_keys(500, seed=31337)
    rng = np.random.default_rng(4)
    for i in range(150, 500, 9):  # revisits
        keys[i] = keys[i - 120] + (rng.integers(-1, 2, 20) / 60.0).astype(np.float32) * (rng.uniform(size=20) < 0.25)
    dummy = np.full(20, 0.5, np.float32)
    db = O.OracleRingDB(dummy=dummy)
    cands = np.full((500, 3), -1, np.int32)
    for i, k in enumerate(keys):
        c = db.query_then_enqueue(k)
        cands[i, : len(c)] = c
    # 50 batched queries against the final index: raw 3-NN (index, squared distance bits)
    q = (keys[rng.integers(500, size=50)] + rng.normal(0, 0.02, (50, 20))).astype(np.float32)
    idx = np.zeros((50, 3), np.int32)
    dist = np.zeros((50, 3), np.float32)
    dbinf = O.OracleRingDB(dummy=dummy, thres=np.inf)
    dbinf.add_points(keys)
    for i in range(50):
        ii, dd = dbinf.knn(q[i])
        idx[i], dist[i] = ii, dd
    np.savez_compressed(os.path.join(HERE, "ringkey_500.npz"), keys=keys, dummy=dummy, candidates=cands, queries=q,
                        knn_idx=idx, knn_dist=dist)


if __name__ == "__main__":
    tracker_fixture()
    ringkey_fixture()
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
