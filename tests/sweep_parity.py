"""Randomised GPU-vs-oracle sweep (not collected by pytest; run on a GPU box: python tests/sweep_parity.py).
Counts the cases where the two paths take different accept / reject / break decisions -- decisions taken on float
sums whose last bits depend on the summation order."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from direct_stereo_slam_amd.tracker import Context
from direct_stereo_slam_amd import synth as S
from _scenes import make_scene, hip_tracker, oracle_tracker
ctx = Context(0)
bad = 0; n = 0; ev_mis = 0; maxd = 0.0
for size, template in (("tiny", "dense"), ("small", "dense"), ("small", "sparse"), ("odd", "dense")):
    for seed in range(100, 160):
        for ms in (1.0, 3.0):
            sc = make_scene(size, seed=seed, template=template, n0=3000, motion_scale=ms)
            orc, trk = oracle_tracker(sc), hip_tracker(ctx, sc)
            go, po, ao, lo, fo = orc.track(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
            gg, pg, ag, lg = trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
            st = ctx.stats()
            n += 1
            same_ev = list(st.evals)[:sc.nl] == orc.eval_counts()[0][:sc.nl]
            d = np.abs(np.asarray(pg) - np.asarray(po)).max()
            if gg != go:
                bad += 1; print("GOOD FLAG MISMATCH", size, template, seed, ms, gg, go)
            elif not same_ev:
                ev_mis += 1; print("eval count mismatch", size, template, seed, ms, list(st.evals)[:sc.nl], orc.eval_counts()[0][:sc.nl], "pose diff", d)
            elif go:
                maxd = max(maxd, d)
            # scale
            eo, so = orc.optimize_scale(1.0, sc.nl - 1)
            eg, sg = trk.optimizeScale(1.0, sc.nl - 1)
            if not (np.isclose(eo, eg, rtol=1e-4, equal_nan=True) and np.isclose(so, sg, rtol=1e-4, equal_nan=True)):
                print("scale mismatch", size, template, seed, ms, (eo, so), (eg, sg))
print("cases", n, "flag mismatches", bad, "eval-count mismatches", ev_mis, "max pose diff (same trajectory)", maxd)
