"""experiment build with shader-clock stamps (tools/experiments/r06_stamps_build.py): runs bench.py's headline leg in this process,
then prints where the tick kernels' cycles went"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DSM_HOTPATH_LIB", os.path.join(ROOT, "scratch", "lib_stamps", "libdsm_hotpath.so"))
import bench  # noqa: E402

sys.argv = ["bench.py", "--quick"] + sys.argv[1:]
a = bench.parse()
bench.bench_tracking(a)
L = C.CDLL(os.environ["DSM_HOTPATH_LIB"])
buf = (C.c_ulonglong * 64)()
assert L.dsm_debug_stamps(buf, 0) == 0
g = list(buf)
out = {"tick_eval_workgroup_cycles_by_level": []}
tot = sum(g[2 * (2 * l + r)] for l in range(6) for r in range(2))
for l in range(6):
    for r in range(2):
        k = 2 * (2 * l + r)
        if g[k + 1]:
            out["tick_eval_workgroup_cycles_by_level"].append({"lvl": l, "residual_only": bool(r), "items": g[k + 1], "cycles_per_item": round(g[k] / g[k + 1]),
                                                               "share_of_workgroup_time": round(g[k] / tot, 4)})
names = ["entry -> first state word (status check)", "state / descriptor / partial loads issued and arrived, psum to LDS", "barrier", "reduce_partials_final",
         "decision, H / b build, state update", "LDLT solve", "extrapolation, SE3 exp, pose product, increment norm", "make_eval (rotation matrix, R K^-1, affine)",
         "wait for the speculative proposal (wave 1)", "end_level", "state write-back", "items of the next tick (tick_push) / retire"]
n = g[48]
out["tick_lm_steps_with_a_proposal"] = n
out["tick_lm_steps_without"] = g[49]
out["tick_lm_phase_cycles"] = {names[i]: round(g[32 + i] / max(1, n)) for i in range(12)}
out["tick_lm_total_cycles"] = round(sum(g[32:44]) / max(1, n))
print(json.dumps(out, indent=1))
