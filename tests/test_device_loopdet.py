"""Row N3, second half, on the device (csrc/loopdet_kernels.hip): the voxel "highest point" filter of
generate_spherical_points (generate_spherical_points.h:44-85) and the polar binning / ring key / signature of
ScanContext::generate (ScanContext.cpp:96-141), batched over keyframes -- against the host forms behind the same C ABI
(bit for bit) and against the numpy/scipy oracle (oracle/scancontext.py; ring key and selection bit-exact, signature to
1e-9: the oracle's PCA sums run in numpy's order)."""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S
from direct_stereo_slam_amd.ringdb import generate_spherical_points, loop_descriptors_batch, scancontext_generate
from oracle import scancontext as SC

pytestmark = pytest.mark.gpu


def make_job(seed, n_kf=12, n_pts=6000, span=55.0):
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(seed)
    kf_ids = np.arange(100, 100 + n_kf)
    rot = rng.normal(0, 0.08, (n_kf, 3))
    if n_kf > 7:
        rot[3] = [0.0, 0.9, 0.0]  # rotated too far against the current keyframe: trimmed
        rot[7] = [0.6, 0.0, 0.1]
    poses = np.hstack([rng.normal(0, 5, (n_kf, 3)), rot])
    Rc = Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix()
    cur_cw = np.hstack([Rc, rng.normal(0, 1, (3, 1))])
    pt_kf = rng.choice(np.concatenate([kf_ids, [999]]), n_pts)  # 999: a keyframe that is not in the map
    # a street canyon in the current camera frame (y down): ground, two walls, clutter
    g = np.stack([rng.uniform(-span, span, n_pts), 1.6 + rng.normal(0, 0.05, n_pts), rng.uniform(-span, span, n_pts)], 1)
    walls = rng.random(n_pts) < 0.35
    g[walls, 0] = np.where(rng.random(walls.sum()) < 0.5, 8.0, -9.0)
    g[walls, 1] = rng.uniform(-5, 1.6, walls.sum())
    xyz = (g - cur_cw[:, 3]) @ Rc  # world coordinates
    xyz[:50] = xyz[0] + rng.normal(0, 0.05, (50, 3))  # a crowded voxel
    xyz[50:60] = xyz[50]  # exact ties: the first index must win
    return kf_ids, poses, cur_cw, pt_kf, xyz


@pytest.mark.parametrize("lidar_range", [40.0, 25.5])
def test_device_pair_equals_host_forms_and_oracle(ctx, lidar_range):
    jobs = [make_job(s, n_pts=n) for s, n in ((1, 6000), (2, 20000), (3, 300), (4, 6000))]
    res = loop_descriptors_batch(ctx, jobs, lidar_range)
    assert len(res) == len(jobs)
    for job, r in zip(jobs, res):
        keep_h, sel_h, pts_h = generate_spherical_points(job[0], job[1], job[2], lidar_range, job[3], job[4])
        np.testing.assert_array_equal(r["kf_keep"], keep_h)
        np.testing.assert_array_equal(r["sel_idx"], sel_h)          # same voxels, same winners, same (ascending voxel) order
        np.testing.assert_array_equal(r["pts_spherical"], pts_h)    # bit for bit
        keep_o, sel_o, pts_o = SC.generate_spherical_points(job[0], job[1], job[2], lidar_range, job[3], job[4])
        np.testing.assert_array_equal(r["sel_idx"], sel_o)
        np.testing.assert_array_equal(r["pts_spherical"], pts_o)
        assert 0 < len(sel_h) < len(job[3])
        rk_h, si_h, sv_h, tfm_h = scancontext_generate(pts_h, lidar_range)
        np.testing.assert_array_equal(r["ringkey"], rk_h)
        np.testing.assert_array_equal(r["sig_idx"], si_h)
        np.testing.assert_array_equal(r["sig_val"], sv_h)           # same sums in the same order
        np.testing.assert_array_equal(r["tfm_pca_rig"], tfm_h)
        rk_o, si_o, sv_o, tfm_o = SC.generate(pts_o, lidar_range)
        np.testing.assert_array_equal(r["ringkey"], rk_o)           # loop-closure keys: bit exact
        np.testing.assert_array_equal(r["sig_idx"], si_o)
        np.testing.assert_allclose(r["sig_val"], sv_o, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["tfm_pca_rig"], tfm_o, atol=1e-9)


def test_point_filter_alone_and_degenerate_jobs(ctx):
    job = make_job(9, n_pts=4000)
    r = loop_descriptors_batch(ctx, [job], 40.0, scancontext=False)[0]
    keep_h, sel_h, pts_h = generate_spherical_points(job[0], job[1], job[2], 40.0, job[3], job[4])
    np.testing.assert_array_equal(r["sel_idx"], sel_h)
    np.testing.assert_array_equal(r["pts_spherical"], pts_h)
    # every keyframe trimmed / no points: empty output, no ScanContext requested
    kf_ids, poses, cur_cw, pt_kf, xyz = job
    r0 = loop_descriptors_batch(ctx, [(kf_ids[:0], poses[:0], cur_cw, pt_kf[:0], xyz[:0])], 40.0, scancontext=False)[0]
    assert len(r0["sel_idx"]) == 0 and r0["pts_spherical"].shape == (0, 3)
    far = xyz + 1000.0  # all beyond lidar_range
    r1 = loop_descriptors_batch(ctx, [(kf_ids, poses, cur_cw, pt_kf, far)], 40.0, scancontext=False)[0]
    assert len(r1["sel_idx"]) == 0
    from direct_stereo_slam_amd._lib import DsmError

    with pytest.raises(DsmError):
        loop_descriptors_batch(ctx, [(kf_ids, poses, cur_cw, pt_kf, far)], 40.0)  # ScanContext of an empty point set
    with pytest.raises(DsmError):
        loop_descriptors_batch(ctx, [job], 500.0)  # dense voxel grid out of range


def test_batch_of_eleven_sequences_feeds_the_ring_key_search(ctx):
    """BASELINE configs[4] shape: eleven concurrent sequences hand one keyframe each to the loop thread; their ring keys go
    straight into the device k-NN"""
    from direct_stereo_slam_amd.ringdb import RingKeyDB

    jobs = [make_job(20 + s, n_pts=5000 + 500 * s) for s in range(11)]
    res = loop_descriptors_batch(ctx, jobs, 40.0)
    keys = np.stack([r["ringkey"] for r in res])
    assert np.all((keys >= 0) & (keys <= 1)) and np.allclose(keys * 60, np.round(keys * 60))
    db = RingKeyDB(ctx, capacity=64)
    db.add_points(keys)
    packed = db.knn_packed_host(keys)
    assert np.array_equal(packed[:, 0] & 0xFFFFFFFF, np.arange(1, 12))  # every key finds itself (index 0 is the dummy slot)


def test_fused_loop_chain_equals_the_sequential_calls(ctx):
    """dsm_loop_detect_batch: descriptors -> ring key -> k-NN -> candidates as ONE enqueue and ONE read-back for a batch of keyframes.
    Semantics = LoopHandler's per-keyframe chain run job after job (LoopHandler.cpp:186,236,247; search_place.h:25-57): query j
    searches the index as it stands after the keys of queries 0 .. j-1 were enqueued -- including the keys those enqueues moved out of
    the delay queue.  Same descriptors, same candidates, same index afterwards, bit for bit, in batches small and large, with a
    short delay margin so that keys mature inside the batches."""
    from direct_stereo_slam_amd.ringdb import RingKeyDB

    rng = np.random.default_rng(7)
    # revisits: every place is described twice (second time from a slightly different cloud), in a shuffled order
    places = [make_job(300 + s, n_pts=3000 + 200 * (s % 5)) for s in range(14)]
    jobs = []
    for rep in range(2):
        for s in rng.permutation(len(places)):
            kf_ids, poses, cur_cw, pt_kf, xyz = places[s]
            jobs.append((kf_ids, poses, cur_cw, pt_kf, xyz + rng.normal(0, 0.003 * rep, xyz.shape)))
    for margin, batches in ((5, (1, 3, 5, 5, 4, 5, 5)), (9, (7, 9, 9, 3))):
        assert sum(batches) == len(jobs)
        seq_db, fus_db = (RingKeyDB(ctx, capacity=64, margin=margin) for _ in range(2))
        seq = []
        for job in jobs:  # the sequential chain: one descriptor call, one search_ringkey call per keyframe
            r = loop_descriptors_batch(ctx, [job], 40.0)[0]
            r["candidates"] = seq_db.search_ringkey(r["ringkey"])
            seq.append(r)
        fus, i = [], 0
        for b in batches:
            fus += loop_descriptors_batch(ctx, jobs[i:i + b], 40.0, db=fus_db, selected_points=(i % 2 == 0))
            i += b
        assert any(r["candidates"] for r in seq), "the scenario must produce loop candidates"
        for a, f in zip(seq, fus):
            assert np.array_equal(a["ringkey"], f["ringkey"]) and a["candidates"] == f["candidates"]
            assert np.array_equal(a["sig_idx"], f["sig_idx"]) and np.array_equal(a["sig_val"], f["sig_val"])
            assert np.array_equal(a["tfm_pca_rig"], f["tfm_pca_rig"]) and a["n_out"] == f["n_out"]
            if "sel_idx" in f:
                assert np.array_equal(a["sel_idx"], f["sel_idx"]) and np.array_equal(a["pts_spherical"], f["pts_spherical"])
        assert seq_db.size() == fus_db.size()
        q = np.stack([r["ringkey"] for r in seq[:8]])
        assert np.array_equal(seq_db.knn_packed_host(q), fus_db.knn_packed_host(q))  # the same index afterwards
    with pytest.raises(Exception):
        loop_descriptors_batch(ctx, jobs[:6], 40.0, db=RingKeyDB(ctx, capacity=64, margin=5))  # more keyframes than the delay margin


@pytest.mark.gpu
def test_clouds_in_page_locked_memory_are_read_in_place(ctx):
    """A caller that keeps its clouds in page-locked memory (dsm_host_alloc) is served without the host staging copy: the device reads
    the buffers where they are (loop_gather_kernel).  Same results as from pageable memory, all jobs page-locked, none, or mixed; the
    keep decision per point (keyframe trimmed or unknown, generate_spherical_points.h:55) is taken on the device in every case."""
    jobs = [make_job(500 + s, n_pts=2500 + 300 * (s % 4)) for s in range(6)]
    kf_ids, poses, cur_cw, pt_kf, xyz = jobs[2]
    pt_kf = pt_kf.copy()
    pt_kf[::7] = 99999  # points of a keyframe that is not in the window: dropped
    jobs[2] = (kf_ids, poses, cur_cw, pt_kf, xyz)
    ref = loop_descriptors_batch(ctx, jobs, 40.0)
    for pinned in (True, [True, False, True, False, False, True]):
        got = loop_descriptors_batch(ctx, jobs, 40.0, pinned_clouds=pinned)
        for a, b in zip(ref, got):
            assert a["n_out"] == b["n_out"] and np.array_equal(a["kf_keep"], b["kf_keep"])
            for k in ("sel_idx", "pts_spherical", "ringkey", "sig_idx", "sig_val", "tfm_pca_rig"):
                assert np.array_equal(a[k], b[k]), k
    # ... and the device's keep decision equals the oracle's (host restatement of the whole chain)
    keep_o, sel_o, pts_o = SC.generate_spherical_points(*jobs[2][:3], 40.0, jobs[2][3], jobs[2][4])
    assert np.array_equal(ref[2]["sel_idx"], sel_o) and np.array_equal(ref[2]["kf_keep"], keep_o)
