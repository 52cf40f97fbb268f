#!/usr/bin/env python3
"""How many of the LM route flips against the oracle are owned by the kernel's off-decision-path shortcuts (hardware reciprocal in
the Huber weight, FMA-contracted gradient interpolation / Jacobian / accumulation, reciprocal-multiply for new_idepth) and how many
by the summation order alone?  (the accept test at TrackerAndScaler.cpp:559.)

  make -C direct_stereo_slam_amd/csrc OUT=../lib_huber EXTRA=-DDSM_IEEE_HUBER      # the IEEE division only
  make -C direct_stereo_slam_amd/csrc OUT=../lib_refops EXTRA=-DDSM_REF_POINT_OPS  # every per-point value in the reference's sequence
  for lib in lib lib_huber lib_refops; do DSM_HOTPATH_LIB=$PWD/direct_stereo_slam_amd/$lib/libdsm_hotpath.so python tools/experiments/lm_flip_attribution.py; done

Runs the 256-scene sweep of tests/test_sweep_parity.py and prints the flips by kind; with DSM_REF_POINT_OPS the only difference
left between device and oracle is the ORDER of the float sums (E, H, b)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _scenes import hip_tracker, make_scene, oracle_tracker  # noqa: E402

from direct_stereo_slam_amd import synth as S  # noqa: E402
from direct_stereo_slam_amd.tracker import Context  # noqa: E402

ctx = Context(0)
n = 0
flips = {"flag": [], "evals": [], "route": [], "scale_evals": []}
worst_same = 0.0
for size, template in (("tiny", "dense"), ("small", "dense"), ("small", "sparse"), ("odd", "dense")):
    for seed in range(100, 132):
        for ms in (1.0, 3.0):
            sc = make_scene(size, seed=seed, template=template, n0=3000, motion_scale=ms)
            orc, trk = oracle_tracker(sc), hip_tracker(ctx, sc)
            go, po, ao, lo, fo = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
            gg, pg, ag, lg = trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
            ev_g, ev_o = list(ctx.stats().evals)[:sc.nl], orc.eval_counts()[0][:sc.nl]
            n += 1
            d = float(np.abs(np.asarray(pg) - np.asarray(po)).max())
            key = (size, template, seed, ms)
            if gg != go:
                flips["flag"].append(key)
            elif ev_g != ev_o:
                flips["evals"].append(key + (d,))
            elif go and d >= 1e-4:
                flips["route"].append(key + (d,))
            elif go:
                worst_same = max(worst_same, d)
            orc.optimize_scale(1.0, sc.nl - 1)
            trk.optimizeScale(1.0, sc.nl - 1)
            if list(ctx.stats().evals)[:sc.nl] != orc.eval_counts()[0][:sc.nl]:
                flips["scale_evals"].append(key)
print(json.dumps({"lib": os.environ.get("DSM_HOTPATH_LIB", "default"), "scenes": n, "flag_flips": len(flips["flag"]), "eval_count_flips": len(flips["evals"]),
                  "route_flips_same_counts": len(flips["route"]), "scale_eval_count_flips": len(flips["scale_evals"]),
                  "max_pose_diff_same_route": worst_same, "cases": {k: [list(map(str, c)) for c in v] for k, v in flips.items()}}))
