"""CPU: the SURVEY.md section 8a rows that stay on the host (A4 makeCoarseDepthL0, A12 ScanContext::generate)
-- product C++ (csrc/host_capi.cpp, reached through the C ABI) against the oracle restatements."""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S
from oracle import oracle as O
from oracle import scancontext as OSC

from _scenes import make_scene


def test_make_coarse_depth_l0_matches_oracle(built):
    from direct_stereo_slam_amd.tracker import make_coarse_depth_l0

    sc = make_scene("small", seed=61)
    rng = np.random.default_rng(61)
    npts = 1500  # ~ the reference's 2000 active points (main.cpp:88-89)
    pu = rng.uniform(3, sc.w - 4, npts).astype(np.float32)
    pv = rng.uniform(3, sc.h - 4, npts).astype(np.float32)
    idl0 = sc.scene.idepth(sc.K, sc.w, sc.h)
    pid = idl0[(pv + 0.5).astype(int), (pu + 0.5).astype(int)] * rng.uniform(0.9, 1.1, npts).astype(np.float32)
    pid[:5] = -0.1  # non-positive idepths must be dropped at the emit stage (:302)
    pw = np.sqrt(1e-3 / (rng.uniform(1e-3, 10, npts) + 1e-12)).astype(np.float32)
    pu[10:14] = pu[10]  # several points on one pixel: weighted mean
    pv[10:14] = pv[10]
    ref = [p.copy() for p in sc.ref_p]
    ref[0][40, 50, 0] = np.nan  # non-finite reference colour must be dropped (:302)
    orc = O.OracleTracker(sc.w, sc.h, sc.nl, sc.T, sc.K)
    orc.make_k(*sc.K)
    exp = orc.make_coarse_depth_l0(pu, pv, pid, pw, ref)
    got = make_coarse_depth_l0(sc.w, sc.h, sc.nl, pu, pv, pid, pw, ref)
    for a_list, b_list in zip(got, exp):
        for a, b in zip(a_list, b_list):
            np.testing.assert_array_equal(a, b)  # same float32 operation order: bit exact
    # template density after dilation: several times the number of active points
    assert len(got[0][0]) > 3 * npts * 0.8
    assert all(len(got[0][l]) > 0 for l in range(sc.nl))
    assert np.all(got[2][0] > 0)


def test_template_from_make_coarse_depth_tracks(built):
    """A4 output is a valid ABI input: the oracle tracker converges from a dilated sparse template"""
    sc = make_scene("small", seed=62)
    rng = np.random.default_rng(62)
    npts = 2000
    pu = rng.uniform(3, sc.w - 4, npts).astype(np.float32)
    pv = rng.uniform(3, sc.h - 4, npts).astype(np.float32)
    idl0 = sc.scene.idepth(sc.K, sc.w, sc.h)
    pid = idl0[(pv + 0.5).astype(int), (pu + 0.5).astype(int)]
    from direct_stereo_slam_amd.tracker import make_coarse_depth_l0

    tpl = make_coarse_depth_l0(sc.w, sc.h, sc.nl, pu, pv, pid, np.ones(npts, np.float32), sc.ref_p)
    orc = O.OracleTracker(sc.w, sc.h, sc.nl, sc.T, sc.K)
    orc.make_k(*sc.K)
    orc.set_ref(0, 0, 0, 1.0, *tpl)
    orc.set_frame(0, sc.new_p, 1.0)
    good, pose, aff, last, _ = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    assert good
    np.testing.assert_allclose(pose[4:], sc.gt_pose[4:], atol=2e-2)


def lidar_like_points(seed, n=4000, lidar_range=40.0):
    rng = np.random.default_rng(seed)
    # a street canyon: ground plane + two walls, in a rotated and translated frame
    g = np.stack([rng.uniform(-35, 35, n // 2), rng.uniform(-30, 30, n // 2), rng.normal(0, 0.05, n // 2)], 1)
    wl = np.stack([rng.uniform(-35, 35, n // 4), np.full(n // 4, 8.0), rng.uniform(0, 6, n // 4)], 1)
    wr = np.stack([rng.uniform(-35, 35, n // 4), np.full(n // 4, -9.0), rng.uniform(0, 4, n // 4)], 1)
    P = np.vstack([g, wl, wr])
    R = S.so3_exp(rng.normal(0, 0.4, 3))
    return P @ R.T + rng.normal(0, 3, 3)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_scancontext_generate_matches_oracle(built, seed):
    from direct_stereo_slam_amd.ringdb import scancontext_generate

    pts = lidar_like_points(seed)
    rk, si, sv, tfm = scancontext_generate(pts, 40.0)
    rk_o, si_o, sv_o, tfm_o = OSC.generate(pts, 40.0)
    np.testing.assert_allclose(tfm, tfm_o, atol=1e-9)
    # binning is a floor of a double expression: points within 1e-9 of a bin edge could flip; none here
    np.testing.assert_array_equal(si, si_o)
    np.testing.assert_array_equal(rk, rk_o)  # ring key: integer counts / 60 in float32 -- bit exact
    np.testing.assert_allclose(sv, sv_o, rtol=1e-9, atol=1e-12)
    # invariants of the descriptor (ScanContext.cpp:123-141)
    assert np.all((rk >= 0) & (rk <= 1)) and np.allclose(rk * 60, np.round(rk * 60))
    for s in range(60):
        v = sv[(si // 20) == s]
        if len(v):
            assert abs((v ** 2).sum() - 1) < 1e-9  # per-sector L2 normalisation
    al = (np.c_[pts, np.ones(len(pts))] @ tfm.T)[:, :3]
    assert al[:, 0].var() <= al[:, 1].var() <= al[:, 2].var()  # x is the smallest-variance ("up") axis
    np.testing.assert_allclose(al.mean(0), 0, atol=1e-9)


def test_ringkey_is_invariant_to_viewpoint_rotation(built):
    """the property the place-recognition stage relies on: a rigid motion of the cloud leaves the
    ring key unchanged (PCA alignment), up to points that cross a bin edge"""
    from direct_stereo_slam_amd.ringdb import scancontext_generate

    pts = lidar_like_points(5)
    rk0 = scancontext_generate(pts, 40.0)[0]
    R = S.so3_exp(np.array([0.3, -0.2, 1.1]))
    rk1 = scancontext_generate(pts @ R.T + np.array([4.0, -2.0, 1.0]), 40.0)[0]
    assert np.abs(rk0 - rk1).max() <= 2 / 60 + 1e-6
    assert ((rk0 - rk1) ** 2).sum() < 0.01  # far below RINGKEY_THRES = 0.1


def test_generate_spherical_points_matches_oracle(built):
    """N3 producer of the ScanContext input (generate_spherical_points.h:27-85): keyframe trimming by orientation, range
    filter, one (highest) point per 1 x 0.5 x 1 m voxel -- C ABI vs the numpy/scipy oracle, then through ScanContext"""
    from scipy.spatial.transform import Rotation

    from direct_stereo_slam_amd.ringdb import generate_spherical_points, scancontext_generate
    from oracle import scancontext as SC

    rng = np.random.default_rng(7)
    n_kf, n_pts, rng_m = 12, 6000, 40.0
    kf_ids = np.arange(100, 100 + n_kf)
    rot = rng.normal(0, 0.08, (n_kf, 3))
    rot[3] = [0.0, 0.9, 0.0]   # rotated too far against the current keyframe: trimmed
    rot[7] = [0.6, 0.0, 0.1]
    poses = np.hstack([rng.normal(0, 5, (n_kf, 3)), rot])
    Rc = Rotation.from_rotvec([0.02, -0.05, 0.01]).as_matrix()
    cur_cw = np.hstack([Rc, rng.normal(0, 1, (3, 1))])
    pt_kf = rng.choice(np.concatenate([kf_ids, [999]]), n_pts)  # 999: a keyframe that is not in the map
    xyz = np.stack([rng.uniform(-55, 55, n_pts), rng.normal(0, 1.5, n_pts), rng.uniform(-55, 55, n_pts)], 1)
    xyz[:50] = xyz[0] + rng.normal(0, 0.05, (50, 3))  # a crowded voxel
    keep_c, sel_c, pts_c = generate_spherical_points(kf_ids, poses, cur_cw, rng_m, pt_kf, xyz)
    keep_o, sel_o, pts_o = SC.generate_spherical_points(kf_ids, poses, cur_cw, rng_m, pt_kf, xyz)
    assert not keep_c[3] and not keep_c[7] and keep_c.sum() == n_kf - 2
    np.testing.assert_array_equal(keep_c, keep_o)
    np.testing.assert_array_equal(sel_c, sel_o)
    np.testing.assert_array_equal(pts_c, pts_o)
    assert 0 < len(sel_c) < n_pts and np.all(np.linalg.norm(pts_c, axis=1) < rng_m)
    assert np.all(keep_c[np.searchsorted(kf_ids, pt_kf[sel_c])])  # only points of kept, known keyframes survive
    # feeds ScanContext::generate: the ring key of the selected points
    rk = scancontext_generate(pts_c, rng_m)[0]
    rk_o = SC.generate(pts_o, rng_m)[0]
    np.testing.assert_array_equal(rk, rk_o)
    # degenerate inputs
    k0, s0, p0 = generate_spherical_points(kf_ids[:0], poses[:0], cur_cw, rng_m, pt_kf[:0], xyz[:0])
    assert len(k0) == 0 and len(s0) == 0 and p0.shape == (0, 3)


def test_trajectory_writer_matches_savepose_format(built, tmp_path):
    """LoopHandler::savePose (LoopHandler.cpp:59-80): `incoming_id x y z`, operator<< of doubles at setprecision(6)"""
    from direct_stereo_slam_amd.ringdb import write_trajectory

    rng = np.random.default_rng(3)
    ids = np.array([0, 3, 4, 9, 1000, 123456])
    t = np.vstack([rng.normal(0, 100, (4, 3)), [[1e-7, -2.5e6, 0.0]], [[1.0, 2.0, 3.0]]])
    f = tmp_path / "dslam.txt"
    write_trajectory(f, ids, t)
    lines = f.read_text().splitlines()
    assert len(lines) == len(ids)
    for line, i, p in zip(lines, ids, t):
        assert line == f"{i} {p[0]:.6g} {p[1]:.6g} {p[2]:.6g}"  # a default C++ stream prints %g-style with 6 significant digits
    assert lines[-1] == "123456 1 2 3" and lines[-2].split()[1:] == ["1e-07", "-2.5e+06", "0"]
