#!/usr/bin/env python3
"""How often does the REFERENCE ALGORITHM (CPU oracle) itself end away from the ground truth on a synthetic scene family, from the
identity guess, at the bench's S2 geometry?  (the bench scenes must not be degenerate.)  CPU only.

    python tools/scene_failure_rate.py <family> [frames] [textures]
family: plane (round 1-3's single textured plane, eight equal sinusoids) | relief (ReliefScene: 1/f texture over a smooth relief)
        | plane1f (the plane with the 1/f texture) ; prints failures (> 1 cm), evaluations per frame and level."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_stereo_slam_amd import synth as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

W, H, NL = 1248, 384, 6
fx, fy, cx, cy = S.KITTI_K_RAW
K = (fx, fy, cx + (1248 - 1241) / 2.0, cy + (384 - 376) / 2.0)


def make_scene(family, seed):
    if family == "plane":
        return S.PlaneScene(seed=seed)
    if family == "plane1f":
        return S.ReliefScene(seed=seed, relief_m=0.0)
    if family.startswith("relief:"):  # relief:<spectrum exponent>[:<relief metres>[:<shortest wavelength px>]]
        a = family.split(":")
        return S.ReliefScene(seed=seed, spectrum=float(a[1]), relief_m=float(a[2]) if len(a) > 2 else 0.25,
                             wavelength_px=(float(a[3]) if len(a) > 3 else 6.0, 480.0))
    return S.ReliefScene(seed=seed)


def run(job):
    family, f, n_tex = job
    k = f % n_tex
    seed = 0x5EED0000 + k
    scene = make_scene(family, seed)
    rng0 = np.random.default_rng(seed)
    ref = scene.render(K, W, H, noise=2.0, rng=rng0)
    rng = np.random.default_rng(0x5EED0000 + 0x100000 * (f // n_tex + 1) + k)
    R, t = S.random_motion(rng)
    new = scene.render(K, W, H, R, t, a=0.02, b=3.0, noise=2.0, rng=rng)
    ref_p = O.make_images(ref, NL, native=True)
    tpl = S.dense_template(scene, K, W, H, NL, ref_p)
    orc = O.OracleTracker(W, H, NL, S.KITTI_T_STEREO, K, O.default_params(native=True), native=True)
    orc.use_sse(True)
    orc.make_k(*K)
    orc.set_ref(0, 0.0, 0.0, 1.0, *tpl)
    orc.set_frame(0, O.make_images(new, NL, native=True), 1.0)
    good, pose, aff, last, flow = orc.track(S.IDENTITY_POSE, [0, 0], NL - 1)
    gt = S.pose_from_Rt(R, t)
    return bool(good), float(np.abs(pose[4:] - gt[4:]).max()), orc.eval_counts()[0][:NL]


if __name__ == "__main__":
    family = sys.argv[1] if len(sys.argv) > 1 else "relief"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    n_tex = int(sys.argv[3]) if len(sys.argv) > 3 else 18
    with ProcessPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        res = list(ex.map(run, [(family, f, n_tex) for f in range(n)]))
    err = np.array([r[1] for r in res])
    ev = np.array([r[2] for r in res], float)
    print(f"{family}: {n} frames, {n_tex} textures: tracked flag false on {sum(not r[0] for r in res)}, translation error > 1 cm on {(err > 0.01).sum()} "
          f"({100.0 * (err > 0.01).mean():.1f} %), > 1 mm on {(err > 0.001).sum()}; median error {np.median(err):.2e} m; "
          f"evaluations per frame by level {np.round(ev.mean(0), 2).tolist()}, max {ev.max(0).astype(int).tolist()}")
    print("failing frames:", [i for i in range(n) if err[i] > 0.01])
