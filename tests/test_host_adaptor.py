"""GPU: the C++ host adaptor (direct_stereo_slam_amd/host/TrackerAndScaler.hpp -- the reference's
class surface on the C ABI) gives the same results as the Python mirror on the same fixture."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from direct_stereo_slam_amd import synth as S

from _scenes import hip_tracker, make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adaptor_matches_python_mirror(ctx, tmp_path):
    sc = make_scene("small", seed=17)
    path = tmp_path / "fixture.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("iii", sc.w, sc.h, sc.nl))
        f.write(np.asarray(sc.K, np.float32).tobytes())
        f.write(np.asarray(sc.T, np.float64).tobytes())
        for l in range(sc.nl):
            f.write(struct.pack("i", len(sc.tpl[0][l])))
            for a in sc.tpl:
                f.write(np.ascontiguousarray(a[l], np.float32).tobytes())
        for pyr in (sc.new_p, sc.right_p):
            for l in range(sc.nl):
                f.write(np.ascontiguousarray(pyr[l], np.float32).tobytes())
    exe = os.path.join(ROOT, "direct_stereo_slam_amd", "host", "_build", "host_adaptor_demo")
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    trk = hip_tracker(ctx, sc)
    good, pose, aff, last = trk.trackNewestCoarse(S.IDENTITY_POSE, [0, 0], sc.nl - 1)
    err, s = trk.optimizeScale(1.0, sc.nl - 1)
    assert bool(res["good"]) == good and res["ref_id"] == 7
    np.testing.assert_array_equal(res["pose"], pose)  # same library, same launches: bit identical
    np.testing.assert_array_equal(res["aff"], aff)
    assert np.float32(res["scale"]) == np.float32(s) and np.float32(res["scale_err"]) == np.float32(err)
    np.testing.assert_allclose(res["flow"], trk.lastFlowIndicators, rtol=1e-7)
