"""Seeded synthetic stereo scenes of KITTI shape for tests and bench.py.

A textured tilted plane is ray-cast from the reference (keyframe) camera, from a new left camera
at a known pose and from the right camera of the stereo rig (cams/kitti/0_2/T_stereo.yaml:4-7),
so the photometric model of the tracker (TrackerAndScaler.cpp:747-793, :1061-1109) holds exactly
up to image noise: pose, affine brightness and stereo scale all have a known ground truth.

Pure numpy; no device code, no oracle code.
"""
import math

import numpy as np

# cams/kitti/0_2/camera0.txt:1-4 ("crop" from 1241x376 to 1232x368 keeps fx,fy and shifts cx,cy by
# half the removed border -- upstream DSO Undistort, crop mode)
KITTI_RAW = (1241, 376)
KITTI_WORK = (1232, 368)
KITTI_K_RAW = (718.856, 718.856, 607.1928, 185.2157)
KITTI_T_STEREO = np.array([[1, 0, 0, -0.5372], [0, 1, 0, 0], [0, 0, 1, 0.000000001], [0, 0, 0, 1]], np.float64)


def kitti_K_work():
    fx, fy, cx, cy = KITTI_K_RAW
    return (fx, fy, cx - (KITTI_RAW[0] - KITTI_WORK[0]) / 2.0, cy - (KITTI_RAW[1] - KITTI_WORK[1]) / 2.0)


def pyramid_levels_dso(w, h, max_levels=6):
    """upstream DSO setGlobalCalib rule (called at main.cpp:151-153): halve while both dims stay even
    and w*h > 5000; at most PYR_LEVELS."""
    n = 1
    while (w % 2 == 0) and (h % 2 == 0) and (w * h > 5000) and n < max_levels:
        w //= 2
        h //= 2
        n += 1
    return n


def level_K(K, lvl):
    """makeK rule, TrackerAndScaler.cpp:126-133 (float32 arithmetic is redone by the tracker itself;
    this float64 version is only for ray casting)."""
    fx, fy, cx, cy = K
    s = float(1 << lvl)
    return (fx / s, fy / s, (cx + 0.5) / s - 0.5, (cy + 0.5) / s - 0.5)


def so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + math.sin(th) / th * W + (1 - math.cos(th)) / th**2 * (W @ W)


def rot_to_quat(R):
    """rotation matrix -> (qx,qy,qz,qw), w >= 0"""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def pose_from_Rt(R, t):
    return np.concatenate([rot_to_quat(R), np.asarray(t, np.float64)])


IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


class PlaneScene:
    """Textured plane n.X = d in the reference camera frame."""

    def __init__(self, seed=0, normal=(-0.5, 0.6, 1.0), dist=6.0, n_waves=8, wavelength_px=(8.0, 128.0),
                 fx_ref=718.856):
        rng = np.random.default_rng(seed)
        n = np.asarray(normal, np.float64)
        self.n = n / np.linalg.norm(n)
        self.d = float(dist)
        a = np.array([1.0, 0, 0]) - self.n * self.n[0]
        self.e1 = a / np.linalg.norm(a)
        self.e2 = np.cross(self.n, self.e1)
        z0 = self.d / self.n[2]
        px = z0 / fx_ref  # metres per pixel at the image centre
        lam = np.exp(rng.uniform(math.log(wavelength_px[0]), math.log(wavelength_px[1]), n_waves)) * px
        ang = rng.uniform(0, 2 * math.pi, n_waves)
        self.freq = np.stack([np.cos(ang) / lam, np.sin(ang) / lam], 1)  # cycles per metre
        self.phase = rng.uniform(0, 2 * math.pi, n_waves)
        self.amp = rng.uniform(0.5, 1.0, n_waves)
        self.amp *= 100.0 / self.amp.sum()  # image in [28, 228]

    def texture(self, X):
        p = X @ self.e1
        q = X @ self.e2
        v = np.full(p.shape, 128.0)
        for k in range(len(self.amp)):
            v += self.amp[k] * np.sin(2 * math.pi * (self.freq[k, 0] * p + self.freq[k, 1] * q) + self.phase[k])
        return v

    def _rays(self, K, w, h):
        fx, fy, cx, cy = K
        xs = (np.arange(w, dtype=np.float64) - cx) / fx
        ys = (np.arange(h, dtype=np.float64) - cy) / fy
        X, Y = np.meshgrid(xs, ys)
        return np.stack([X, Y, np.ones_like(X)], -1)  # (h,w,3)

    def render(self, K, w, h, R=None, t=None, a=0.0, b=0.0, noise=0.0, rng=None):
        """image seen by a camera with x_cam = R x_ref + t ; brightness exp(a)*I + b ; + N(0,noise^2)."""
        R = np.eye(3) if R is None else R
        t = np.zeros(3) if t is None else np.asarray(t, np.float64)
        r = self._rays(K, w, h)
        nR = R @ self.n  # n . R^T r = (R n) . r
        z = (self.d + self.n @ (R.T @ t)) / (r @ nR)
        Xc = r * z[..., None]
        Xref = (Xc - t) @ R  # R^T (Xc - t)
        img = math.exp(a) * self.texture(Xref) + b
        if noise > 0:
            img = img + (rng or np.random.default_rng(0)).normal(0, noise, img.shape)
        return img.astype(np.float32)

    def idepth(self, K, w, h, R=None, t=None):
        """inverse depth seen by a camera with x_cam = R x_ref + t (default: the reference camera)"""
        r = self._rays(K, w, h)
        if R is None:
            return ((r @ self.n) / self.d).astype(np.float32)
        t = np.asarray(t, np.float64)
        return ((r @ (R @ self.n)) / (self.d + self.n @ (R.T @ t))).astype(np.float32)


class ReliefScene(PlaneScene):
    """A textured RELIEF over the plane n.X = d: the surface point above plane coordinates (p, q) is raised by hgt(p, q) along the
    normal -- a few smooth bumps of 3 - 9 m wavelength, 0.15 m in all: depth structure and parallax instead of one homography, slopes
    below the shallowest viewing angle so that no ray meets the surface twice -- and the
    texture is broadband: 32 sinusoids of wavelengths 6 ... 480 px (amplitude ~ wavelength^spectrum; 0 = equal weights, 1 = a
    natural-image-like 1/f spectrum) instead of eight between 8 and 128 px, so that every pyramid level sees unambiguous structure
    and a frame's cost surface has one basin at the scale of the motion prior (the eight-wave single-plane scenes drive the
    reference algorithm itself into a wrong minimum on 6-11 % of the frames from the identity guess; on this family
    the CPU oracle tracks 96 of 96 frames: tools/scene_failure_rate.py).  Ray casting: Newton iterations from the plane's
    intersection."""

    def __init__(self, seed=0, normal=(-0.5, 0.6, 1.0), dist=6.0, n_waves=32, wavelength_px=(6.0, 480.0), fx_ref=718.856, relief_m=0.15,
                 relief_wavelength_m=(3.0, 9.0), n_bumps=5, spectrum=0.0):
        super().__init__(seed, normal, dist, n_waves, wavelength_px, fx_ref)
        rng = np.random.default_rng(seed ^ 0x0BADC0DE)
        z0 = self.d / self.n[2]
        px = z0 / fx_ref
        lam = np.exp(rng.uniform(math.log(wavelength_px[0]), math.log(wavelength_px[1]), n_waves)) * px
        ang = rng.uniform(0, 2 * math.pi, n_waves)
        self.freq = np.stack([np.cos(ang) / lam, np.sin(ang) / lam], 1)
        self.phase = rng.uniform(0, 2 * math.pi, n_waves)
        self.amp = rng.uniform(0.6, 1.0, n_waves) * (lam / lam.max()) ** spectrum
        self.amp *= 100.0 / np.sqrt((self.amp ** 2).sum()) / 2.2  # rms ~ 32 grey levels; clipped to [0, 255] when rendered
        bl = rng.uniform(relief_wavelength_m[0], relief_wavelength_m[1], n_bumps)
        ba = rng.uniform(0, 2 * math.pi, n_bumps)
        self.bfreq = np.stack([np.cos(ba) / bl, np.sin(ba) / bl], 1)
        self.bphase = rng.uniform(0, 2 * math.pi, n_bumps)
        self.bamp = rng.uniform(0.5, 1.0, n_bumps)
        self.bamp *= relief_m / self.bamp.sum()

    def height(self, p, q, grad=False):
        hgt = np.zeros(np.shape(p))
        gp, gq = np.zeros(np.shape(p)), np.zeros(np.shape(p))
        for k in range(len(self.bamp)):
            ph = 2 * math.pi * (self.bfreq[k, 0] * p + self.bfreq[k, 1] * q) + self.bphase[k]
            hgt = hgt + self.bamp[k] * np.sin(ph)
            if grad:
                c = self.bamp[k] * 2 * math.pi * np.cos(ph)
                gp, gq = gp + c * self.bfreq[k, 0], gq + c * self.bfreq[k, 1]
        return (hgt, gp, gq) if grad else hgt

    def texture(self, X):
        return np.clip(super().texture(X), 0.0, 255.0)

    def _cast(self, K, w, h, R, t):
        """depth z along every pixel's ray (x, y, 1) of a camera x_cam = R x_ref + t, and the surface points in the reference frame"""
        R = np.eye(3) if R is None else R
        t = np.zeros(3) if t is None else np.asarray(t, np.float64)
        r = self._rays(K, w, h)
        dn, d1, d2 = r @ (R @ self.n), r @ (R @ self.e1), r @ (R @ self.e2)  # d(n.X, p, q) / dz along the ray
        z = (self.d + self.n @ (R.T @ t)) / dn  # the plane's intersection
        if self.bamp.sum() > 0:
            for _ in range(5):  # Newton on f(z) = n.X(z) - d - hgt(p(z), q(z)); quadratic convergence (slopes << 1)
                Xref = (r * z[..., None] - t) @ R
                hgt, gp, gq = self.height(Xref @ self.e1, Xref @ self.e2, grad=True)
                z = z - (Xref @ self.n - self.d - hgt) / (dn - gp * d1 - gq * d2)
        return z, (r * z[..., None] - t) @ R

    def render(self, K, w, h, R=None, t=None, a=0.0, b=0.0, noise=0.0, rng=None):
        _, Xref = self._cast(K, w, h, R, t)
        Xp = Xref - np.multiply.outer(self.height(Xref @ self.e1, Xref @ self.e2), self.n)  # the texture lives on the plane coordinates
        img = math.exp(a) * self.texture(Xp) + b
        if noise > 0:
            img = img + (rng or np.random.default_rng(0)).normal(0, noise, img.shape)
        return img.astype(np.float32)

    def idepth(self, K, w, h, R=None, t=None):
        z, _ = self._cast(K, w, h, R, t)
        return (1.0 / z).astype(np.float32)


def dense_template(scene, K, w, h, nlevels, ref_pyr, idepth_scale=1.0, R=None, t=None):
    """every interior pixel 2<=x<wl-2, 2<=y<hl-2 is a template point (the emit rule of
    makeCoarseDepthL0, TrackerAndScaler.cpp:291-314), row-major order as the reference emits.
    (R, t): pose of the keyframe camera in the scene's reference frame (default identity)."""
    us, vs, ids, cs = [], [], [], []
    for l in range(nlevels):
        wl, hl = w >> l, h >> l
        idl = scene.idepth(level_K(K, l), wl, hl, R, t) * np.float32(idepth_scale)
        ys, xs = np.mgrid[2:max(2, hl - 2), 2:max(2, wl - 2)]  # (levels smaller than 5 x 5 have no interior)
        us.append(xs.astype(np.float32).ravel())
        vs.append(ys.astype(np.float32).ravel())
        ids.append(np.ascontiguousarray(idl[2:max(2, hl - 2), 2:max(2, wl - 2)]).ravel())
        cs.append(np.ascontiguousarray(ref_pyr[l][2:max(2, hl - 2), 2:max(2, wl - 2), 0]).ravel())
    return us, vs, ids, cs


def sparse_template(scene, K, w, h, nlevels, ref_pyr, n0=10000, seed=1, idepth_scale=1.0):
    """replay-like template: n0 random interior pixels at level 0, n0/4^l (>=64) at level l,
    sorted row-major like the reference's emit order."""
    rng = np.random.default_rng(seed)
    us, vs, ids, cs = dense_template(scene, K, w, h, nlevels, ref_pyr, idepth_scale)
    out = [[], [], [], []]
    for l in range(nlevels):
        n = len(us[l])
        k = min(n, max(64, n0 >> (2 * l)))
        sel = np.sort(rng.choice(n, k, replace=False))
        for o, a in zip(out, (us, vs, ids, cs)):
            o.append(np.ascontiguousarray(a[l][sel]))
    return out


def random_motion(rng, sigma_t=(0.05, 0.05, 0.2), sigma_r=0.005):
    t = rng.normal(0, 1, 3) * np.asarray(sigma_t)
    w = rng.normal(0, sigma_r, 3)
    return so3_exp(w), t
