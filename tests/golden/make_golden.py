#!/usr/bin/env python3
"""Mints the golden fixtures under tests/golden/ from the CPU oracle (oracle/dsm_oracle.c, parity
build).  The reference itself has no tests, golden vectors or buildable sources in this image
(SURVEY.md sections 4 and 8c), so these vectors pin the ORACLE (against accidental change) and the
HIP path (against the oracle) -- they are not outputs of the reference binary.

    python tests/golden/make_golden.py        # rewrites every *.npz of this directory

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

from _scenes import make_affine_scene, make_relief_frames, make_scene, oracle_tracker  # noqa: E402
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


def tracker_small_fixture():
    """the 308x92 pair of SURVEY.md section 8c (three levels): inputs (raw images; the pyramids are makeImages of them),
    dense template, expected track, the scale results of the front end's guess list (FrontEnd.cpp:995-1003) and of the
    fixed benchmark schedule"""
    sc = make_scene("small", seed=2025)
    orc = oracle_tracker(sc)
    out = {"w": sc.w, "h": sc.h, "nl": sc.nl, "K": np.asarray(sc.K, np.float64), "T": sc.T, "gt_pose": sc.gt_pose,
           "ref_img": sc.ref_img, "new_img": sc.new_img, "right_img": sc.right_img}
    for l in range(sc.nl):
        for name, arr in zip(("u", "v", "id", "c"), sc.tpl):
            out[f"tpl_{name}{l}"] = arr[l]
    good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    out["track_good"], out["track_pose"], out["track_aff"], out["track_last"], out["track_flow"] = good, pose, aff, last, flow
    out["track_evals"] = np.array(orc.eval_counts()[0])
    guesses = np.array([0.1, 1, 5, 10, 15, 25, 30, 50], np.float32)  # FrontEnd.cpp:995 scale_guess
    res, sev = [], []
    for g in guesses:
        res.append(orc.optimize_scale(float(g), sc.nl - 1))
        sev.append(orc.eval_counts()[0])
    out["scale_guesses"] = guesses
    out["scale_evals"] = np.array(sev)
    out["scale_err"] = np.array([r[0] for r in res], np.float32)
    out["scale_out"] = np.array([r[1] for r in res], np.float32)
    op = O.default_params()
    op.fixed_schedule = 3
    orc3 = oracle_tracker(sc, op)
    g3, p3, a3, l3, _ = orc3.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    out["fixed3_pose"], out["fixed3_aff"], out["fixed3_last"] = p3, a3, l3
    np.savez_compressed(os.path.join(HERE, "tracker_small.npz"), **out)


def pose_estimator_fixture():
    """PoseEstimator::estimate (PoseEstimator.cpp:84-506) on 1500 loop-closure points of a 308x92 keyframe pair"""
    from test_pose_estimator import gt_matrix, loop_inputs

    sc = make_scene("small", seed=2026, a=0.01, b=2.0)
    xyz, cols = loop_inputs(sc, n=1500, seed=3)
    pe = O.OraclePoseEstimator(sc.w, sc.h, sc.nl)
    out = {"w": sc.w, "h": sc.h, "nl": sc.nl, "K": np.asarray(sc.K, np.float64), "new_img": sc.new_img, "xyz": xyz, "gt": gt_matrix(sc)}
    for l in range(sc.nl):
        out[f"col{l}"] = cols[l]
    for tag, guess in (("eye", np.eye(4)), ("far", np.array([[1, 0, 0, 3.0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]))):
        ok, T, err, inl = pe.estimate(xyz, cols, 1.0, sc.new_p, 1.0, sc.K, sc.nl - 1, guess)
        out[f"{tag}_guess"], out[f"{tag}_ok"], out[f"{tag}_T"], out[f"{tag}_err"], out[f"{tag}_inl"] = guess, ok, T, np.float32(err), inl
    np.savez_compressed(os.path.join(HERE, "pose_estimator_small.npz"), **out)


def loop_descriptor_fixture():
    """generate_spherical_points (generate_spherical_points.h:27-85) + ScanContext::generate (ScanContext.cpp:19-141) of one
    keyframe job: the selected points, ring key, sparse signature and PCA transform (numpy / scipy oracle)"""
    from oracle import scancontext as SC
    from test_device_loopdet import make_job

    kf_ids, poses, cur_cw, pt_kf, xyz = make_job(2027, n_kf=12, n_pts=3000)
    keep, sel, pts = SC.generate_spherical_points(kf_ids, poses, cur_cw, 40.0, pt_kf, xyz)
    rk, si, sv, tfm = SC.generate(pts, 40.0)
    np.savez_compressed(os.path.join(HERE, "loop_descriptor.npz"), kf_ids=kf_ids, poses=poses, cur_cw=cur_cw, pt_kf=pt_kf, xyz=xyz,
                        lidar_range=40.0, kf_keep=keep, sel_idx=sel, pts_spherical=pts, ringkey=rk, sig_idx=si, sig_val=sv, tfm_pca_rig=tfm)


def tracker_relief_fixture():
    """the bench's default scene family (synth.ReliefScene) at 308x92x3: four frames of one texture handed over as camera
    bytes (mono8), dense template; expected track per frame and the first frame's scale optimisation.  Replayed through the
    streaming form (tests/test_stream.py) and against the oracle itself (tests/test_golden.py)."""
    frames = make_relief_frames("small", 4, 1, seed0=0x5EED0200, u8=True)
    sc = frames[0]
    out = {"w": sc.w, "h": sc.h, "nl": sc.nl, "K": np.asarray(sc.K, np.float64), "T": sc.T, "n_frames": len(frames),
           "ref_u8": sc.ref_img.astype(np.uint8), "right_u8": sc.right_img.astype(np.uint8),
           "new_u8": np.stack([f.new_img.astype(np.uint8) for f in frames]), "gt_pose": np.stack([f.gt_pose for f in frames])}
    for l in range(sc.nl):
        for name, arr in zip(("u", "v", "id", "c"), sc.tpl):
            out[f"tpl_{name}{l}"] = arr[l]
    res = []
    for i, f in enumerate(frames):
        orc = oracle_tracker(f)
        good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, [0, 0], f.nl - 1)
        res.append((good, pose, aff, last, flow, np.array(orc.eval_counts()[0])))
        if i == 0:
            err, s = orc.optimize_scale(1.0, f.nl - 1)
            out["scale_err"], out["scale_out"], out["scale_evals"] = np.float32(err), np.float32(s), np.array(orc.eval_counts()[0])
    for j, name in enumerate(("track_good", "track_pose", "track_aff", "track_last", "track_flow", "track_evals")):
        out[name] = np.stack([np.asarray(r[j]) for r in res])
    np.savez_compressed(os.path.join(HERE, "tracker_relief_small.npz"), **out)


def tracker_affine_fixture():
    """the photometric branch every real keyframe takes: reference affine (-0.3, 12), exposures 0.8 -> 1.3 on a
    308x92x3 relief frame; plus the same frame with a ZERO reference exposure (fromToVecExposure's "either 0 => both 1").  Expected: the
    fused evaluation of every level at the ground truth (rs, H, b, warped count) and the track from the keyframe's own affine."""
    out = {}
    for tag, ref_exp in (("exp", 0.8), ("zero", 0.0)):
        sc = make_affine_scene("small", seed=41, ref_aff=(-0.3, 12.0), ref_exposure=ref_exp, new_exposure=1.3, new_aff=(-0.25, 20.0), family="relief")
        orc = oracle_tracker(sc)
        if tag == "exp":
            out.update({"w": sc.w, "h": sc.h, "nl": sc.nl, "K": np.asarray(sc.K, np.float64), "T": sc.T, "gt_pose": sc.gt_pose, "gt_aff": sc.gt_aff,
                        "ref_aff": np.array(sc.ref_aff), "new_exposure": sc.new_exposure})
            for l in range(sc.nl):
                for name, arr in zip(("u", "v", "id", "c"), sc.tpl):
                    out[f"tpl_{name}{l}"] = arr[l]
        out[f"{tag}_ref_exposure"] = ref_exp
        out[f"{tag}_new_img"] = sc.new_img
        for lvl in range(sc.nl):
            rs = orc.calc_res_pose(lvl, sc.gt_pose, sc.gt_aff, 20.0)
            out[f"{tag}_rs{lvl}"], out[f"{tag}_E64_{lvl}"] = rs, orc.last_energy_f64()
            H, b = orc.calc_gs_pose(lvl, sc.gt_pose, sc.gt_aff)
            out[f"{tag}_H{lvl}"], out[f"{tag}_b{lvl}"], out[f"{tag}_n{lvl}"] = H, b, orc.pose_warped_n()
        good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, list(sc.ref_aff), sc.nl - 1)
        out[f"{tag}_track_good"], out[f"{tag}_track_pose"], out[f"{tag}_track_aff"], out[f"{tag}_track_last"], out[f"{tag}_track_flow"] = good, pose, aff, last, flow
        out[f"{tag}_track_evals"] = np.array(orc.eval_counts()[0])
    np.savez_compressed(os.path.join(HERE, "tracker_affine_small.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "relief":  # (adds the round-5 fixture without rewriting the others)
        tracker_relief_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "affine":  # (round 6)
        tracker_affine_fixture()
        sys.exit(0)
    tracker_fixture()
    tracker_relief_fixture()
    tracker_affine_fixture()
    ringkey_fixture()
    tracker_small_fixture()
    pose_estimator_fixture()
    loop_descriptor_fixture()
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
