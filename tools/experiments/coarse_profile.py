#!/usr/bin/env python3
"""Developer aid: phase breakdown (shader cycles of workgroup 0) of coarse_kernel from a library built with
-DDSM_COARSE_PROFILE:  python tools/experiments/coarse_profile.py [bench args...]  (runs bench.one_step on a tiny batch)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import argparse

import numpy as np

import bench

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
from direct_stereo_slam_amd import _lib
from direct_stereo_slam_amd.tracker import Context

lib = _lib.load()
raw = C.CDLL(os.path.join(ROOT, "direct_stereo_slam_amd", "lib", "libdsm_hotpath.so"))
ctx = Context(0)
ctx.set_streams(args.streams)
wl = bench.build_workload(args, ctx, args.config)
B = len(wl["trackers"])
kf = list(range(0, B, args.kf_every))
for _ in range(2):
    bench.one_step(ctx, wl, kf)
buf = (C.c_longlong * 16)()
ebuf = (C.c_longlong * 16)()
raw.dsm_debug_coarse_prof(buf, 1)
raw.dsm_debug_eval_prof(ebuf, 1)
N = 5
for _ in range(N):
    bench.one_step(ctx, wl, kf)
raw.dsm_debug_coarse_prof(buf, 1)
v = list(buf)
names = ["turn-around", "inputs+staging", "evaluation", "partial reduction", "LM step (wave 0)", "barrier after step"]
rounds = max(1, v[8])
print(f"workgroup 0 over {N} steps: {v[8]} LM rounds ({v[9]} with a speculative candidate), {v[10] / rounds:.1f} chunks per round")
for i, n in enumerate(names):
    print(f"  {n:22s} {v[i] / rounds:9.0f} cycles per round")
print(f"  {'total':22s} {sum(v[:6]) / rounds:9.0f} cycles per round")
calls = max(1, v[15])
print(f"propose_pose, {v[15]} calls (coarse_kernel and lm_kernel, workgroup 0): LDLT {v[11] / calls:.0f}, SE3 exp {v[12] / calls:.0f}, "
      f"product / norms {v[13] / calls:.0f}, next evaluation's inputs {v[14] / calls:.0f} cycles")
raw.dsm_debug_eval_prof(ebuf, 1)
e = list(ebuf)
ne = max(1, e[8])
print(f"eval_kernel<pose>, levels >= 1, problem 0 chunk 0: {e[8]} workgroups, {e[9] / ne:.1f} points per thread: entry -> inputs {e[0] / ne:.0f}, "
      f"-> first template entries {e[1] / ne:.0f}, per-point loop {e[2] / ne:.0f}, reduction + store {e[3] / ne:.0f} cycles")
