"""numpy restatement ("oracle") of ScanContext::generate, TEST INFRASTRUCTURE ONLY.
Follows src/loop_closure/loop_detection/ScanContext.cpp:19-66 (align_points_PCA) and :78-141
(generate).  Eigen's SelfAdjointEigenSolver is replaced by numpy.linalg.eigh (LAPACK); both return
ascending eigenvalues, the eigenvector signs are not defined by either -- the same orientation
convention as the product is applied (largest-magnitude component positive).  PARITY UNPINNED
(no reference binary, no reference tests)."""
import math

import numpy as np


def align_points_pca(pts):
    pts = np.asarray(pts, np.float64)
    mean = pts.sum(0) / len(pts)                      # :22-29
    mat = pts - mean                                  # :32-37
    cov = mat.T @ mat                                 # :40
    w, V = np.linalg.eigh(cov)                        # :41-44 (ascending)
    for c in range(3):
        if V[np.argmax(np.abs(V[:, c])), c] < 0:
            V[:, c] = -V[:, c]
    aligned = mat @ V                                 # :47-54
    tfm = np.eye(4)
    tfm[:3, :3] = V.T                                 # :57-60
    tfm[:3, 3] = -V.T @ mean                          # :61-64
    return aligned, tfm


def generate(pts, lidar_range, num_s=60, num_r=20):
    aligned, tfm = align_points_pca(pts)
    ringkey = np.zeros(num_r, np.float32)
    max_height = np.full(num_s * num_r, -lidar_range - 1.0)
    for x, y, z in aligned:                           # :96-119 (x: up, polar in (y,z))
        rho = math.sqrt(y * y + z * z)
        theta = math.atan2(z, y)
        while theta < 0:
            theta += 2.0 * math.pi
        while theta >= 2.0 * math.pi:
            theta -= 2.0 * math.pi
        si = int(theta / (2.0 * math.pi) * num_s)
        ri = int(rho / lidar_range * num_r)
        if ri >= num_r or si >= num_s:
            continue
        max_height[si * num_r + ri] = max(max_height[si * num_r + ri], x)
    idx = np.nonzero(max_height >= -lidar_range)[0]   # :123-131
    for i in idx:
        ringkey[i % num_r] += np.float32(1.0)
    val = max_height[idx].copy()
    norm = np.zeros(num_s)
    for i, v in zip(idx, val):
        norm[i // num_r] += v * v
    ringkey = ringkey / np.float32(num_s)             # :134-136
    norm = np.sqrt(norm)
    val = val / norm[idx // num_r]                    # :139-141
    return ringkey.astype(np.float32), idx.astype(np.int32), val, tfm


def generate_spherical_points(kf_ids, kf_pose_wc, cur_cw, lidar_range, pt_kf_id, pt_xyz):
    """generate_spherical_points.h:27-85 restated with numpy/scipy (TEST INFRASTRUCTURE).  cur_cw: 3x4 camera<-world.
    Keyframes rotated by more than 0.5 rad against the current one are trimmed (:33-41); points of trimmed / unknown
    keyframes and points at or beyond lidar_range are dropped (:55-64); per voxel of 1 x 0.5 x 1 m (RES_X/Y/Z :23-25) the
    highest point (smallest y) is kept, first one on ties (:73-76).  Output order: ascending voxel index (the reference's
    unordered_map order is implementation defined)."""
    from scipy.spatial.transform import Rotation

    cur_cw = np.asarray(cur_cw, np.float64).reshape(3, 4)
    poses = np.asarray(kf_pose_wc, np.float64).reshape(-1, 6)
    kf_keep = np.array([np.linalg.norm(Rotation.from_matrix(cur_cw[:, :3] @ Rotation.from_rotvec(poses[i, 3:]).as_matrix()).as_rotvec()) <= 0.5
                        for i in range(len(kf_ids))], bool)
    keep = {}
    for i, k in enumerate(kf_ids):
        keep[int(k)] = keep.get(int(k), False) or bool(kf_keep[i])
    steps = (1.0, 2.0, 1.0)
    vs0 = int(np.floor(2 * lidar_range * steps[0])) + 1
    vs1 = int(np.floor(2 * lidar_range * steps[1])) + 1
    best = {}
    for i, (k, g) in enumerate(zip(pt_kf_id, np.asarray(pt_xyz, np.float64).reshape(-1, 3))):
        if not keep.get(int(k), False):
            continue
        p = np.array([((cur_cw[r, 0] * g[0] + cur_cw[r, 1] * g[1]) + cur_cw[r, 2] * g[2]) + cur_cw[r, 3] for r in range(3)])
        if np.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) >= lidar_range:
            continue
        xi, yi, zi = (int(np.floor((p[a] + lidar_range) * steps[a])) for a in range(3))
        loc = xi + yi * vs0 + zi * vs0 * vs1
        if loc not in best or p[1] < best[loc][1][1]:
            best[loc] = (i, p)
    locs = sorted(best)
    return kf_keep, np.array([best[l][0] for l in locs], np.int32), np.array([best[l][1] for l in locs]).reshape(-1, 3)
