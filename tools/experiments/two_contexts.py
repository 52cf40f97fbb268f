#!/usr/bin/env python3
"""Experiment (round 3): do two host threads, each driving its own dsm_context over HALF of the frames, overlap one half's
latency-bound small-level phase with the other half's bandwidth-bound level-0/1 phase?
  python tools/experiments/two_contexts.py [bench args...] [--offset-ms X]
Prints frames/s for one context x B frames and for two contexts x B/2 frames (thread 2 started offset-ms later)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import bench

off = 4.0
if "--offset-ms" in sys.argv:
    i = sys.argv.index("--offset-ms")
    off = float(sys.argv[i + 1])
    del sys.argv[i:i + 2]
args = bench.parse()
from direct_stereo_slam_amd.tracker import Context

B = args.batch
steps = args.steps


def run(nctx, offset_ms):
    args.batch = B // nctx
    args.scenes = B  # the same B distinct frames, dealt out to the contexts
    ctxs, wls = [], []
    for c in range(nctx):
        ctx = Context(0)
        ctx.set_streams(args.streams)
        ctxs.append(ctx)
        bench._FRAME_OFFSET = c * (B // nctx)
        wls.append(bench.build_workload(args, ctx, args.config))
    bench._FRAME_OFFSET = 0
    kf = list(range(0, args.batch, args.kf_every))
    for c in range(nctx):
        for _ in range(2):
            bench.one_step(ctxs[c], wls[c], kf)
    bar = threading.Barrier(nctx + 1)

    def worker(c):
        bar.wait()
        if c:
            time.sleep(1e-3 * offset_ms * c)
        for _ in range(steps):
            bench.one_step(ctxs[c], wls[c], kf)
        ctxs[c].sync()

    th = [threading.Thread(target=worker, args=(c,)) for c in range(nctx)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return B * steps / dt, 1e3 * dt / steps


for nctx, o in ((1, 0.0), (2, 0.0), (2, off), (4, off / 2), (1, 0.0), (2, off), (4, off / 2)):
    v, ms = run(nctx, o)
    print(f"contexts {nctx} x {B // nctx} frames, offset {o:4.1f} ms: {v:9.0f} frames/s  {ms:7.3f} ms per step of {B} frames", flush=True)
