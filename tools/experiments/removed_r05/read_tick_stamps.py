import sys, os, runpy, ctypes
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, R)
sys.argv = ["bench.py", "--no-cpu", "--no-second-leg", "--no-fixed-leg", "--no-replay-leg", "--no-ringkey-leg", "--detail-out", "/tmp/d.json"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
except SystemExit:
    pass
from direct_stereo_slam_amd import _lib
L = _lib.load()
out = (ctypes.c_double * 48)()
rc = L.dsm_exp_read_stamps(out)
names = ["count", "consts", "tpl_in", "loop_end", "rows_written", "barrier", "end"]
print("rc", rc, "(mean shader cycles since workgroup entry, wave 0, full evaluations of pose problems; clock ~2.1-2.4 GHz)")
for l in range(6):
    v = [out[l * 8 + k] for k in range(7)]
    if v[0] == 0: continue
    d = [v[1]] + [v[k] - v[k - 1] for k in range(2, 7)]
    print("level", l, "n", int(v[0]), " ".join("%s %.0f" % (names[k], v[k]) for k in range(1, 7)))
    print("        phases: consts %.0f | template wait %.0f | loop %.0f | flow+rowsums %.0f | barrier %.0f | final %.0f" % tuple(d))
