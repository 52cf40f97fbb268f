"""GPU: the C++ host adaptor (direct_stereo_slam_amd/host/TrackerAndScaler.hpp -- the reference's
class surface on the C ABI) gives the same results as the Python mirror on the same fixture."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S

from _scenes import hip_tracker, make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_fixture(sc, path):
    with open(path, "wb") as f:
        f.write(struct.pack("iii", sc.w, sc.h, sc.nl))
        f.write(np.asarray(sc.K, np.float32).tobytes())
        f.write(np.asarray(sc.T, np.float64).tobytes())
        for l in range(sc.nl):
            f.write(struct.pack("i", len(sc.tpl[0][l])))
            for a in sc.tpl:
                f.write(np.ascontiguousarray(a[l], np.float32).tobytes())
        for pyr in (sc.new_p, sc.right_p):
            for l in range(sc.nl):
                f.write(np.ascontiguousarray(pyr[l], np.float32).tobytes())


def run_host(exe_name, path, *args):
    exe = os.path.join(ROOT, "direct_stereo_slam_amd", "host", "_build", exe_name)
    out = subprocess.run([exe, str(path), *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().splitlines()[-1]


def test_reference_binding_through_standin_types_equals_the_plain_adaptor(ctx, tmp_path):
    """host/ReferenceBinding.hpp -- the bodies a maintainer puts behind dso::TrackerAndScaler's public methods, a template over
    the reference's types -- instantiated with stand-ins that carry the member names it touches on Sophus::SE3, dso::AffLight,
    Eigen vectors, dso::FrameHessian and dso::CalibHessian (host/reference_binding_check.cpp), driven like FrontEnd.cpp: the
    FrameHessian -> dIp / exposure / shell id hand-over, the SE3 and AffLight round trips and the Vec5 residuals must give, bit
    for bit, what the plain adaptor gives on the same fixture"""
    sc = make_scene("small", seed=23)
    path = tmp_path / "fixture.bin"
    write_fixture(sc, path)
    plain, bound = run_host("host_adaptor_demo", path, "2"), run_host("reference_binding_check", path)  # (the binding tracks one frame at a time: chunk table 2; its chained small levels are scheduling only)
    a, b = json.loads(plain), json.loads(bound)
    assert a["good"] == 1
    a.pop("stream_results_equal")  # (the plain demo also exercises dsm_host::Stream)
    assert a == b and plain.split(', "stream_results_equal"')[0] == bound.rstrip("}")


def test_cpp_adaptor_matches_python_mirror(ctx, tmp_path):
    sc = make_scene("small", seed=17)
    path = tmp_path / "fixture.bin"
    write_fixture(sc, path)
    res = json.loads(run_host("host_adaptor_demo", path))
    trk = hip_tracker(ctx, sc)
    good, pose, aff, last = trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    err, s = trk.optimizeScale(1.0, sc.nl - 1)
    assert bool(res["good"]) == good and res["ref_id"] == 7
    assert res["stream_results_equal"] == 2  # dsm_host::Stream returned the track AND the scale result bit for bit
    np.testing.assert_array_equal(res["pose"], pose)  # same library, same launches: bit identical
    np.testing.assert_array_equal(res["aff"], aff)
    assert np.float32(res["scale"]) == np.float32(s) and np.float32(res["scale_err"]) == np.float32(err)
    np.testing.assert_allclose(res["flow"], trk.lastFlowIndicators, rtol=1e-7)


def test_cpp_loop_detection_adaptor_runs_the_loop_handler_sequence(ctx, tmp_path):
    """direct_stereo_slam_amd/host/LoopDetection.hpp (search_place.h / ScanContext.h / generate_spherical_points.h surface in
    C++ on the C ABI) driven like LoopHandler::run: 130 keyframes, the last 25 revisit the first ones -> ring-key candidates
    after the LOOP_MARGIN delay, ScanContext match; every step equal to the Python mirror of the same ABI"""
    from direct_stereo_slam_amd.ringdb import RingKeyDB, generate_spherical_points, scancontext_generate
    from direct_stereo_slam_amd._lib import load
    from test_device_loopdet import make_job

    n_frames, rng_m = 130, 40.0
    jobs = []
    for fr in range(n_frames):
        src = fr - 105 if fr >= 105 else fr  # frames 105.. see the places of frames 0.. again
        kf_ids, poses, cur_cw, pt_kf, xyz = make_job(500 + src, n_kf=6, n_pts=1500)
        if fr >= 105:  # the same place, a slightly different view and a little noise
            xyz = xyz + np.random.default_rng(fr).normal(0, 0.02, xyz.shape)
        jobs.append((kf_ids, poses, cur_cw, pt_kf, xyz))
    path = tmp_path / "loop.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("i", n_frames) + struct.pack("d", rng_m))
        for kf_ids, poses, cur_cw, pt_kf, xyz in jobs:
            f.write(struct.pack("ii", len(kf_ids), len(pt_kf)))
            f.write(np.asarray(kf_ids, np.int32).tobytes() + np.asarray(poses, np.float64).tobytes() + np.asarray(cur_cw, np.float64).tobytes())
            f.write(np.asarray(pt_kf, np.int32).tobytes() + np.asarray(xyz, np.float64).tobytes())
    exe = os.path.join(ROOT, "direct_stereo_slam_amd", "host", "_build", "loop_adaptor_demo")
    traj = tmp_path / "dslam.txt"
    out = subprocess.run([exe, str(path), str(traj)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(res) == n_frames and len(traj.read_text().splitlines()) == n_frames
    L = load()
    db = RingKeyDB(ctx)
    sigs, n_match = [], 0
    for fr, (job, r) in enumerate(zip(jobs, res)):
        keep, sel, pts = generate_spherical_points(job[0], job[1], job[2], rng_m, job[3], job[4])
        rk, si, sv, _ = scancontext_generate(pts, rng_m)
        cand = db.search_ringkey(rk)
        assert r["n_sel"] == len(sel) and r["n_kf_kept"] == int(keep.sum()) and r["n_sig"] == len(si)
        np.testing.assert_array_equal(np.asarray(r["ringkey"], np.float32), rk)
        assert r["candidates"] == cand and r["index_size"] == db.size()
        if cand:
            best, bd = cand[0], np.float32(1.1)
            for c in cand:
                ci, cv = sigs[c]
                d = np.float32(L.dsm_sc_distance(si.ctypes.data_as(_ip()), sv.ctypes.data_as(_dp()), len(si), ci.ctypes.data_as(_ip()),
                                                 cv.ctypes.data_as(_dp()), len(ci), 60))
                if bd > d:
                    best, bd = c, d
            assert r["matched"] == best and np.float32(r["diff"]) == bd
            n_match += r["matched"] == fr - 105
        else:
            assert r["matched"] == -1
        sigs.append((np.ascontiguousarray(si, np.int32), np.ascontiguousarray(sv, np.float64)))
    assert n_match >= 20  # the revisits are found (the first LOOP_MARGIN frames are not yet in the index when they are queried)


def _ip():
    import ctypes as C

    return C.POINTER(C.c_int)


def _dp():
    import ctypes as C

    return C.POINTER(C.c_double)
