"""one keyframe per dsm_loop_detect_batch call, 200 calls (for a rocprofv3 kernel trace of the single-sequence loop chain)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from direct_stereo_slam_amd.tracker import Context
from direct_stereo_slam_amd.ringdb import RingKeyDB, LoopBatch

def job(seed, n_pts=16000):
    rng = np.random.default_rng(seed)
    n_kf = 8
    kf_ids = np.arange(100, 100 + n_kf)
    poses = np.hstack([rng.normal(0, 5, (n_kf, 3)), rng.normal(0, 0.08, (n_kf, 3))])
    cur_cw = np.hstack([np.eye(3), rng.normal(0, 1, (3, 1))])
    pt_kf = rng.choice(kf_ids, n_pts)
    g = np.stack([rng.uniform(-55, 55, n_pts), 1.6 + rng.normal(0, 0.05, n_pts), rng.uniform(-55, 55, n_pts)], 1)
    walls = rng.random(n_pts) < 0.35
    g[walls, 0] = np.where(rng.random(walls.sum()) < 0.5, 8.0, -9.0)
    g[walls, 1] = rng.uniform(-5, 1.6, walls.sum())
    return kf_ids, poses, cur_cw, pt_kf, g - cur_cw[:, 3]

ctx = Context(0)
db = RingKeyDB(ctx, capacity=1 << 16)
one = LoopBatch(ctx, [job(900)], 40.0, db=db, selected_points=False)
for _ in range(20):
    one.run()
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(n):
    one.run()
print("ms per keyframe:", 1e3 * (time.perf_counter() - t0) / n)
