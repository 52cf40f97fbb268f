"""Shared seeded scene builder for the tests (numpy only + the oracle's makeImages)."""
import numpy as np

from direct_stereo_slam_amd import synth as S
from oracle import oracle as O

# (w, h, kitti level whose intrinsics are used, nlevels)
SIZES = {
    "tiny": (154, 46, 3, 2),     # golden fixture size suggested by SURVEY.md section 8c
    "small": (308, 92, 2, 3),
    "medium": (616, 184, 1, 4),
    "kitti": (1232, 368, 0, 5),  # S1: what the reference actually runs
    "odd": (240, 135, 2, 3),     # S3's floor-halved tail: 240x135 -> 120x67 -> 60x33 (odd sizes drop the last row/column)
    # the other BASELINE.json configs' working sizes (crop keeps fx, fy and shifts cx, cy by half the removed border):
    "kitti04": (1216, 368, 0, 5),  # configs[0]: cams/kitti/4_12/camera0.txt:1-4 (1226x370 cropped)
    "malaga": (1024, 768, 0, 5),   # configs[2]: cams/malaga/camera0.txt:1-4
    "kitti6": (1248, 384, 0, 6),   # S2: the metric's own configuration, 1241x376 padded to a multiple of 32, six levels
    "hd6": (1920, 1080, 0, 6),     # S3 / configs[3]: 1920x1080, six floor-halved levels (1920x1080 ... 60x33), full size
    "mini4": (64, 64, 3, 4),       # the smallest pyramid the library accepts: an 8 x 8 coarsest level
}
# (fx, fy, cx, cy) at the working size and the stereo baseline for sizes that are not KITTI-00
CAMERAS = {
    "kitti04": ((707.0912, 707.0912, 601.8873 - (1226 - 1216) / 2.0, 183.1104 - (370 - 368) / 2.0), -0.5372),
    "malaga": ((795.11588, 795.11588, 517.12973, 395.59665), -0.119471),  # cams/malaga/T_stereo.yaml:4-7
    "kitti6": ((718.856, 718.856, 607.1928 + (1248 - 1241) / 2.0, 185.2157 + (384 - 376) / 2.0), -0.5372),
    "hd6": ((718.856 * 1920.0 / 1241.0, 718.856 * 1920.0 / 1241.0, 959.5, 539.5), -0.5372),  # KITTI's field of view
}


class Scene:
    pass


def make_scene(size="small", seed=3, noise=1.0, template="dense", idepth_scale=1.0, a=0.02, b=3.0, n0=10000,
               motion_scale=1.0, wavelength_px=(8.0, 128.0)):
    w, h, klvl, nl = SIZES[size]
    K = S.level_K(S.kitti_K_work(), klvl)
    T = S.KITTI_T_STEREO
    if size in CAMERAS:
        K, baseline = CAMERAS[size]
        T = S.KITTI_T_STEREO.copy()
        T[0, 3] = baseline
    scene = S.PlaneScene(seed=seed, fx_ref=K[0], wavelength_px=wavelength_px)
    rng = np.random.default_rng(seed + 100)
    sc = Scene()
    sc.w, sc.h, sc.nl, sc.K, sc.T = w, h, nl, K, T
    ref = scene.render(K, w, h, noise=noise, rng=rng)
    R, t = S.random_motion(rng)
    if motion_scale != 1.0:
        R, t = S.random_motion(np.random.default_rng(seed + 7), sigma_t=np.array((0.05, 0.05, 0.2)) * motion_scale,
                               sigma_r=0.005 * motion_scale)
    new = scene.render(K, w, h, R, t, a=a, b=b, noise=noise, rng=rng)
    right = scene.render(K, w, h, sc.T[:3, :3], sc.T[:3, 3], noise=noise, rng=rng)
    sc.ref_img, sc.new_img, sc.right_img = ref, new, right
    sc.ref_p, sc.new_p, sc.right_p = (O.make_images(im, nl) for im in (ref, new, right))
    if template == "dense":
        sc.tpl = S.dense_template(scene, K, w, h, nl, sc.ref_p, idepth_scale)
    else:
        sc.tpl = S.sparse_template(scene, K, w, h, nl, sc.ref_p, n0=n0, seed=seed, idepth_scale=idepth_scale)
    sc.gt_pose = S.pose_from_Rt(R, t)
    sc.gt_aff = np.array([a, b])
    sc.scene = scene
    return sc


def regrad(level):
    """Gradient channels of one pyramid level [h, w, 3] rebuilt from channel 0 exactly as FrameHessian::makeImages (upstream
    DSO) forms them: central differences on the flat index for idx in [w, w (h - 1)), zero elsewhere and where the
    difference is not finite.  Tests that edit intensities after make_images call this so that the (I, dx, dy) texels
    stay what the reference would have built -- dsm_tracker_upload_frame checks that they are."""
    h, w, _ = level.shape
    out = np.zeros_like(level)
    I = level[..., 0].reshape(-1).astype(np.float32)
    out[..., 0] = level[..., 0]
    idx = np.arange(w, w * (h - 1))
    with np.errstate(invalid="ignore", over="ignore"):
        dx = np.float32(0.5) * (I[idx + 1] - I[idx - 1])
        dy = np.float32(0.5) * (I[idx + w] - I[idx - w])
    dx[~np.isfinite(dx)] = 0
    dy[~np.isfinite(dy)] = 0
    out.reshape(-1, 3)[idx, 1] = dx
    out.reshape(-1, 3)[idx, 2] = dy
    return out


def _photometry(sc):
    """(reference affine g2l, reference exposure, new-frame exposure) of a scene: (0, 0), 1, 1 unless make_affine_scene set them"""
    return tuple(getattr(sc, "ref_aff", (0.0, 0.0))), float(getattr(sc, "ref_exposure", 1.0)), float(getattr(sc, "new_exposure", 1.0))


def oracle_tracker(sc, params=None):
    ref_aff, ref_exp, new_exp = _photometry(sc)
    orc = O.OracleTracker(sc.w, sc.h, sc.nl, sc.T, sc.K, params)
    orc.make_k(*sc.K)
    orc.set_ref(0, ref_aff[0], ref_aff[1], ref_exp, *sc.tpl)
    orc.set_frame(0, sc.new_p, new_exp)
    orc.set_frame(1, sc.right_p, 1.0)
    return orc


def hip_tracker(ctx, sc, params=None):
    from direct_stereo_slam_amd.tracker import TrackerAndScaler

    ref_aff, ref_exp, new_exp = _photometry(sc)
    trk = TrackerAndScaler(ctx, sc.w, sc.h, sc.nl, sc.T, sc.K, params)
    trk.makeK(*sc.K)
    trk.setCoarseTrackingRef(0, ref_aff, ref_exp, *sc.tpl)
    trk.upload_frame(0, sc.new_p, new_exp)
    trk.upload_frame(1, sc.right_p, 1.0)
    return trk


def aff_from_to(exp_f, exp_t, g2f, g2t):
    """AffLight::fromToVecExposure (upstream DSO; call sites TrackerAndScaler.cpp:647-649,717-720): the brightness map
    I_to = a I_from + b between two frames with affine parameters g2f / g2t and exposures exp_f / exp_t; either exposure 0 => both 1"""
    if exp_f == 0 or exp_t == 0:
        exp_f = exp_t = 1.0
    a = np.exp(g2t[0] - g2f[0]) * exp_t / exp_f
    return a, g2t[1] - a * g2f[1]


def make_affine_scene(size="small", seed=3, ref_aff=(-0.3, 12.0), ref_exposure=0.8, new_exposure=1.3, new_aff=(-0.25, 20.0), family="plane",
                      noise=1.0):
    """A scene whose keyframe has a NONZERO affine brightness (a, b) and whose two frames have different exposure times -- what every
    keyframe of a real sequence has (the reference's photometric model: TrackerAndScaler.cpp:647-649, 673-676, 717-720, 615-626).  The new
    image is rendered with exactly the brightness map the model predicts for the ground-truth new-frame affine `new_aff`, so that the
    tracker converges to (gt_pose, new_aff) from the keyframe's own affine as the initial guess (what FrontEnd hands over)."""
    import math

    sc = make_scene(size, seed=seed, noise=noise) if family == "plane" else make_relief_frames(size, 1, 1, seed0=0x5EED0300 + seed, noise=noise)[0]
    a, b = aff_from_to(ref_exposure, new_exposure, ref_aff, new_aff)
    rng = np.random.default_rng(seed + 500)
    R, t = S.quat_to_rot(sc.gt_pose[:4]), np.asarray(sc.gt_pose[4:], np.float64)
    sc.new_img = sc.scene.render(sc.K, sc.w, sc.h, R, t, a=math.log(a), b=b, noise=noise, rng=rng)
    sc.new_p = O.make_images(sc.new_img, sc.nl)
    sc.ref_aff, sc.ref_exposure, sc.new_exposure = tuple(ref_aff), ref_exposure, new_exposure
    sc.gt_aff = np.array(new_aff, np.float64)
    sc.aff_ll = (a, b)
    return sc


def make_relief_frames(size, n_frames, n_tex, seed0=0x5EED0000, noise=2.0, u8=False):
    """Frames of the bench's default scene family (synth.ReliefScene: a smooth relief under a broadband texture -- bench.py:
    build_frames): `n_tex` textures, each with its keyframe image, dense template and right image; frame f = texture
    f mod n_tex under its own ground-truth motion and noise.  Returns a list of Scene objects (one per frame; frames of a
    texture share the template arrays)."""
    w, h, klvl, nl = SIZES[size]
    K = S.level_K(S.kitti_K_work(), klvl)
    T = S.KITTI_T_STEREO
    if size in CAMERAS:
        K, baseline = CAMERAS[size]
        T = S.KITTI_T_STEREO.copy()
        T[0, 3] = baseline
    q = (lambda im: np.clip(np.rint(im), 0, 255).astype(np.float32)) if u8 else (lambda im: im)
    tex = []
    for k in range(n_tex):
        scene = S.ReliefScene(seed=seed0 + k, fx_ref=K[0])
        rng = np.random.default_rng(seed0 + k)
        ref = q(scene.render(K, w, h, noise=noise, rng=rng))
        right = q(scene.render(K, w, h, T[:3, :3], T[:3, 3], noise=noise, rng=rng))
        ref_p, right_p = O.make_images(ref, nl), O.make_images(right, nl)
        tex.append((scene, ref, right, ref_p, right_p, S.dense_template(scene, K, w, h, nl, ref_p)))
    out = []
    for f in range(n_frames):
        scene, ref, right, ref_p, right_p, tpl = tex[f % n_tex]
        rng = np.random.default_rng(seed0 + 0x100000 * (1 + f // n_tex) + f % n_tex)
        R, t = S.random_motion(rng)
        sc = Scene()
        sc.w, sc.h, sc.nl, sc.K, sc.T = w, h, nl, K, T
        sc.ref_img, sc.right_img = ref, right
        sc.new_img = q(scene.render(K, w, h, R, t, a=0.02, b=3.0, noise=noise, rng=rng))
        sc.ref_p, sc.right_p, sc.new_p = ref_p, right_p, O.make_images(sc.new_img, nl)
        sc.tpl, sc.gt_pose, sc.gt_aff, sc.scene, sc.texture = tpl, S.pose_from_Rt(R, t), np.array([0.02, 3.0]), scene, f % n_tex
        out.append(sc)
    return out
