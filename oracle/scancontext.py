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
