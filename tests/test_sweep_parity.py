"""Randomised GPU-vs-oracle sweep: 256 seeded scenes (four sizes, dense and sparse templates, ground-truth motion x1 and x3).

Every LM decision -- accept / reject, small-step break, cut-off repeat, abort -- is taken on float sums whose last bits
depend on the summation order; the device reduces in a fixed tree, the reference sequentially, so
the two paths may legitimately part ways where a test is decided by those bits.  What must hold:
  * the tracked / aborted FLAG is the same in every case;
  * the two paths take the same route -- same per-level evaluation counts AND poses within 1e-4 -- in at least 96 % of the
    cases (round 2 measured 475 of 480 on the counts; round 4: 8 of these 256 part ways, and a build whose per-point values
    follow the reference's operation sequence to the letter -- IEEE divisions, no FMA: only the ORDER of the sums left --
    parts ways in 7: the summation order owns the flips, tools/experiments/lm_flip_attribution_r04.log); with equal counts
    the scales agree to 1e-4 relative;
  * where they part ways (a test decided by the last bits of a float sum; on the 154x46 scenes with three times the motion
    the valley is flat enough for that to move the end point) both still end within 3 cm of each other.  (Which cases part ways
    follows the summation tree: rounds 2-5 measured at most 1.2-2.0 cm over their parted cases; round 6's chunk table -- one chunk per
    small level -- parts other cases, the farthest pair 2.2 cm apart.  The bar is a sanity bound on "the same basin", not a parity
    figure: the parity figures are the flags, the 96 % and the 1e-4 of the cases that take the oracle's route.)"""
import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S

from _scenes import hip_tracker, make_scene, oracle_tracker

pytestmark = pytest.mark.gpu


def test_sweep_of_random_scenes_takes_the_oracles_decisions(ctx):
    n = flag_mis = ev_mis = 0
    worst_same, worst_diff = 0.0, 0.0
    notes, worst_note = [], None
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
                if gg != go:
                    flag_mis += 1
                    notes.append(("flag", size, template, seed, ms, gg, go))
                elif ev_g != ev_o:
                    ev_mis += 1
                    notes.append(("evals", size, template, seed, ms, ev_g, ev_o, d))
                    if go and d > worst_diff:
                        worst_diff, worst_note = d, notes[-1]
                elif go and d >= 1e-4:
                    ev_mis += 1  # same counts, another route (an accept and a reject swapped places)
                    notes.append(("route", size, template, seed, ms, d))
                    if d > worst_diff:
                        worst_diff, worst_note = d, notes[-1]
                elif go:
                    worst_same = max(worst_same, d)
                eo, so = orc.optimize_scale(1.0, sc.nl - 1)
                eg, sg = trk.optimizeScale(1.0, sc.nl - 1)
                if list(ctx.stats().evals)[:sc.nl] == orc.eval_counts()[0][:sc.nl]:
                    assert np.isclose(so, sg, rtol=1e-4, equal_nan=True) and np.isclose(eo, eg, rtol=1e-3, equal_nan=True), (size, template, seed, ms, (eo, so), (eg, sg))
    assert n == 256
    assert flag_mis == 0, notes
    assert ev_mis <= 0.04 * n, notes
    assert worst_same < 1e-4, worst_same
    print("sweep parity: parted cases", ev_mis, "of", n, "farthest pair", worst_note)
    assert worst_diff < 3e-2, worst_note
