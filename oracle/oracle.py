"""ctypes loader for the CPU oracle (oracle/dsm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (direct_stereo_slam_amd) never does.  Parity is UNPINNED: see dsm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 6

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Params(C.Structure):
    _fields_ = [
        ("huber_th", C.c_float),
        ("coarse_cutoff_th", C.c_float),
        ("scale_xi_rot", C.c_float),
        ("scale_xi_trans", C.c_float),
        ("scale_a", C.c_float),
        ("scale_b", C.c_float),
        ("affine_opt_mode_a", C.c_float),
        ("affine_opt_mode_b", C.c_float),
        ("lambda_extrapolation_limit", C.c_float),
        ("max_iterations", C.c_int * MAX_LEVELS),
        ("fixed_schedule", C.c_int),
    ]


def build(native=False):
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    name = "libdsm_oracle_native.so" if native else "libdsm_oracle.so"
    return os.path.join(_HERE, "_build", name)


_libs = {}


def lib(native=False):
    if native in _libs:
        return _libs[native]
    name = "libdsm_oracle_native.so" if native else "libdsm_oracle.so"
    path = os.path.join(_HERE, "_build", name)
    if native or not os.path.exists(path):
        path = build(native)  # native build is always (re)made on the machine that runs it
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_params_default.argtypes = [C.POINTER(Params)]
    L.orc_tracker_create.restype = vp
    L.orc_tracker_create.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_float_p, C.POINTER(Params)]
    L.orc_tracker_destroy.argtypes = [vp]
    L.orc_tracker_make_k.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
    pp = C.POINTER(c_float_p)
    L.orc_tracker_set_ref.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_float, c_int_p, pp, pp, pp, pp]
    L.orc_tracker_scale_depth.argtypes = [vp, C.c_float]
    L.orc_tracker_get_template.argtypes = [vp, C.c_int, c_int_p, c_float_p, c_float_p, c_float_p, c_float_p]
    L.orc_tracker_set_frame.argtypes = [vp, C.c_int, pp, C.c_float]
    L.orc_calc_res_pose.argtypes = [vp, C.c_int, c_double_p, c_double_p, C.c_float, c_double_p]
    L.orc_calc_gs_pose.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.orc_last_energy_f64.argtypes = [vp]
    L.orc_last_energy_f64.restype = C.c_double
    L.orc_pose_warped_n.argtypes = [vp]
    L.orc_tracker_use_sse.argtypes = [vp, C.c_int]
    L.orc_pose_warped_n.restype = C.c_int
    L.orc_track.argtypes = [vp, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p]
    L.orc_track.restype = C.c_int
    L.orc_get_eval_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.orc_calc_res_scale.argtypes = [vp, C.c_int, C.c_float, C.c_float, c_double_p]
    L.orc_calc_gs_scale.argtypes = [vp, C.c_int, C.c_float, c_float_p, c_float_p]
    L.orc_scale_warped_n.argtypes = [vp]
    L.orc_scale_warped_n.restype = C.c_int
    L.orc_optimize_scale.argtypes = [vp, c_float_p, C.c_int]
    L.orc_optimize_scale.restype = C.c_float
    L.orc_make_images.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int, pp]
    L.orc_make_coarse_depth_l0.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, pp,
                                           c_int_p, pp, pp, pp, pp]
    L.orc_pe_create.restype = vp
    L.orc_pe_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Params)]
    L.orc_pe_destroy.argtypes = [vp]
    L.orc_pe_estimate.argtypes = [vp, C.c_int, c_double_p, pp, C.c_float, pp, C.c_float, c_float_p, C.c_int, c_double_p,
                                  c_float_p, c_int_p]
    L.orc_pe_estimate.restype = C.c_int
    L.orc_se3_exp.argtypes = [c_double_p, c_double_p]
    L.orc_se3_mul.argtypes = [c_double_p, c_double_p, c_double_p]
    L.orc_quat_to_rot.argtypes = [c_double_p, c_double_p]
    L.orc_se3_from_matrix.argtypes = [c_double_p, c_double_p]
    L.orc_ldlt_solve.argtypes = [C.c_int, c_double_p, c_double_p, c_double_p]
    L.orc_ringdb_create.restype = vp
    L.orc_ringdb_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, c_float_p]
    L.orc_ringdb_destroy.argtypes = [vp]
    L.orc_ringdb_size.argtypes = [vp]
    L.orc_ringdb_size.restype = C.c_int64
    L.orc_ringdb_add_points.argtypes = [vp, c_float_p, C.c_int64]
    L.orc_ringdb_query_then_enqueue.argtypes = [vp, c_float_p, c_int_p, c_int_p]
    L.orc_ringdb_knn.argtypes = [vp, c_float_p, c_int_p, c_float_p]
    L.orc_l2_sq.argtypes = [c_float_p, c_float_p, C.c_int]
    L.orc_l2_sq.restype = C.c_float
    L.orc_sc_distance.argtypes = [c_int_p, c_double_p, C.c_int, c_int_p, c_double_p, C.c_int, C.c_int]
    L.orc_sc_distance.restype = C.c_float
    _libs[native] = L
    return L


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ptr_array(arrs):
    arr = (c_float_p * len(arrs))()
    for i, a in enumerate(arrs):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        arr[i] = _fp(a)
    return arr


def default_params(native=False):
    p = Params()
    lib(native).orc_params_default(C.byref(p))
    return p


def make_images(image, nlevels, native=False):
    """upstream DSO makeImages: float image (h,w) -> list of (h_l, w_l, 3) float32 arrays."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    h, w = image.shape
    out = [np.zeros((h >> l, w >> l, 3), np.float32) for l in range(nlevels)]
    lib(native).orc_make_images(_fp(image), w, h, nlevels, _ptr_array(out))
    return out


class OracleTracker:
    """Mirror of TrackerAndScaler (TrackerAndScaler.h:34-137) on the CPU oracle."""

    def __init__(self, w, h, nlevels, T_f1_f0, K1, params=None, native=False):
        self.L = lib(native)
        self.w, self.h, self.nlevels = w, h, nlevels
        self.params = params if params is not None else default_params(native)
        T = np.ascontiguousarray(np.asarray(T_f1_f0, np.float64).reshape(16))
        K1 = np.ascontiguousarray(np.asarray(K1, np.float32))
        self.h_ = self.L.orc_tracker_create(w, h, nlevels, _dp(T), _fp(K1), C.byref(self.params))
        self._keep = {}

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_tracker_destroy(self.h_)
            self.h_ = None

    def use_sse(self, on=True):
        """calcGSSSE* in the SSE-intrinsics form (dsm_oracle_sse.c): the reference's own form, the timed CPU baseline"""
        self.L.orc_tracker_use_sse(self.h_, 1 if on else 0)

    def make_k(self, fx, fy, cx, cy):
        self.L.orc_tracker_make_k(self.h_, fx, fy, cx, cy)

    def set_ref(self, ref_id, ref_a, ref_b, ref_exposure, pc_u, pc_v, pc_idepth, pc_color):
        n = (C.c_int * self.nlevels)(*[len(a) for a in pc_u])
        for lvl, a in enumerate(pc_u):  # the reference's per-level arrays hold w_l * h_l entries (TrackerAndScaler.cpp:52-64)
            if len(a) > (self.w >> lvl) * (self.h >> lvl):
                raise ValueError(f"level {lvl}: more template points than pixels")
        arrs = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (pc_u, pc_v, pc_idepth, pc_color)]
        self.L.orc_tracker_set_ref(self.h_, ref_id, ref_a, ref_b, ref_exposure, n, *[_ptr_array(a) for a in arrs])

    def scale_depth(self, s):
        self.L.orc_tracker_scale_depth(self.h_, s)

    def get_template(self, lvl):
        n = C.c_int()
        self.L.orc_tracker_get_template(self.h_, lvl, C.byref(n), None, None, None, None)
        out = [np.zeros(n.value, np.float32) for _ in range(4)]
        self.L.orc_tracker_get_template(self.h_, lvl, C.byref(n), *[_fp(a) for a in out])
        return out

    def set_frame(self, slot, dIp, exposure=1.0):
        dIp = [np.ascontiguousarray(a, np.float32) for a in dIp]
        self._keep[slot] = dIp  # borrowed pointers, as in the reference
        self.L.orc_tracker_set_frame(self.h_, slot, _ptr_array(dIp), exposure)

    def calc_res_pose(self, lvl, pose, aff, cutoff):
        pose = np.ascontiguousarray(pose, np.float64)
        aff = np.ascontiguousarray(aff, np.float64)
        rs = np.zeros(6)
        self.L.orc_calc_res_pose(self.h_, lvl, _dp(pose), _dp(aff), cutoff, _dp(rs))
        return rs

    def calc_gs_pose(self, lvl, pose, aff):
        pose = np.ascontiguousarray(pose, np.float64)
        aff = np.ascontiguousarray(aff, np.float64)
        H = np.zeros(64)
        b = np.zeros(8)
        self.L.orc_calc_gs_pose(self.h_, lvl, _dp(pose), _dp(aff), _dp(H), _dp(b))
        return H.reshape(8, 8), b

    def last_energy_f64(self):
        return self.L.orc_last_energy_f64(self.h_)

    def pose_warped_n(self):
        return self.L.orc_pose_warped_n(self.h_)

    def track(self, pose, aff, coarsest, min_res=None):
        pose = np.array(pose, np.float64)
        aff = np.array(aff, np.float64)
        mr = np.full(MAX_LEVELS, np.nan) if min_res is None else np.ascontiguousarray(min_res, np.float64)
        last = np.zeros(MAX_LEVELS)
        flow = np.zeros(3)
        good = self.L.orc_track(self.h_, _dp(pose), _dp(aff), coarsest, _dp(mr), _dp(last), _dp(flow))
        return bool(good), pose, aff, last, flow

    def eval_counts(self):
        r = (C.c_int64 * MAX_LEVELS)()
        g = (C.c_int64 * MAX_LEVELS)()
        self.L.orc_get_eval_counts(self.h_, r, g)
        return list(r), list(g)

    def calc_res_scale(self, lvl, scale, cutoff):
        rs = np.zeros(6)
        self.L.orc_calc_res_scale(self.h_, lvl, scale, cutoff, _dp(rs))
        return rs

    def calc_gs_scale(self, lvl, scale):
        H = C.c_float()
        b = C.c_float()
        self.L.orc_calc_gs_scale(self.h_, lvl, scale, C.byref(H), C.byref(b))
        return H.value, b.value

    def scale_warped_n(self):
        return self.L.orc_scale_warped_n(self.h_)

    def optimize_scale(self, scale, coarsest):
        s = C.c_float(scale)
        err = self.L.orc_optimize_scale(self.h_, C.byref(s), coarsest)
        return err, s.value

    def make_coarse_depth_l0(self, pu, pv, pidepth, pweight, ref_dIp):
        pu, pv, pidepth, pweight = [np.ascontiguousarray(a, np.float32) for a in (pu, pv, pidepth, pweight)]
        ref = [np.ascontiguousarray(a, np.float32) for a in ref_dIp]
        n_out = (C.c_int * self.nlevels)()
        outs = [[np.zeros((self.w >> l) * (self.h >> l), np.float32) for l in range(self.nlevels)] for _ in range(4)]
        self.L.orc_make_coarse_depth_l0(self.h_, len(pu), _fp(pu), _fp(pv), _fp(pidepth), _fp(pweight),
                                        _ptr_array(ref), n_out, *[_ptr_array(o) for o in outs])
        return [[o[l][: n_out[l]] for l in range(self.nlevels)] for o in outs]


class OracleRingDB:
    def __init__(self, dim=20, margin=100, k=3, thres=0.1, dummy=None, native=False):
        self.L = lib(native)
        self.dim, self.k = dim, k
        d = None if dummy is None else _fp(np.ascontiguousarray(dummy, np.float32))
        self.h_ = self.L.orc_ringdb_create(dim, margin, k, thres, d)

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_ringdb_destroy(self.h_)
            self.h_ = None

    def size(self):
        return self.L.orc_ringdb_size(self.h_)

    def add_points(self, keys):
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, self.dim)
        self.L.orc_ringdb_add_points(self.h_, _fp(keys), keys.shape[0])

    def query_then_enqueue(self, key):
        key = np.ascontiguousarray(key, np.float32)
        cand = (C.c_int * self.k)()
        nc = C.c_int()
        self.L.orc_ringdb_query_then_enqueue(self.h_, _fp(key), cand, C.byref(nc))
        return [cand[i] for i in range(nc.value)]

    def knn(self, key):
        key = np.ascontiguousarray(key, np.float32)
        idx = (C.c_int * self.k)()
        dist = (C.c_float * self.k)()
        self.L.orc_ringdb_knn(self.h_, _fp(key), idx, dist)
        return list(idx), list(dist)


def sc_distance(a_idx, a_val, b_idx, b_val, sc_width=60, native=False):
    a_idx = np.ascontiguousarray(a_idx, np.int32)
    b_idx = np.ascontiguousarray(b_idx, np.int32)
    a_val = np.ascontiguousarray(a_val, np.float64)
    b_val = np.ascontiguousarray(b_val, np.float64)
    return lib(native).orc_sc_distance(a_idx.ctypes.data_as(c_int_p), _dp(a_val), len(a_idx),
                                       b_idx.ctypes.data_as(c_int_p), _dp(b_val), len(b_idx), sc_width)


class OraclePoseEstimator:
    """PoseEstimator (PoseEstimator.h:34-83) on the CPU oracle"""

    def __init__(self, w, h, nlevels, params=None, native=False):
        self.L = lib(native)
        self.params = params if params is not None else default_params(native)
        self.nlevels = nlevels
        self.h_ = self.L.orc_pe_create(w, h, nlevels, C.byref(self.params))

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_pe_destroy(self.h_)
            self.h_ = None

    def estimate(self, xyz, colors, ref_ab_exposure, new_dIp, new_ab_exposure, new_cam, coarsest_lvl, ref_to_new):
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        colors = [np.ascontiguousarray(c, np.float32) for c in colors]
        new_dIp = [np.ascontiguousarray(a, np.float32) for a in new_dIp]
        cam = np.ascontiguousarray(new_cam, np.float32)
        T = np.ascontiguousarray(ref_to_new, np.float64).reshape(16).copy()
        err, inl = C.c_float(), C.c_int()
        ok = self.L.orc_pe_estimate(self.h_, len(xyz), _dp(xyz), _ptr_array(colors), ref_ab_exposure, _ptr_array(new_dIp),
                                    new_ab_exposure, _fp(cam), coarsest_lvl, _dp(T), C.byref(err), C.byref(inl))
        return bool(ok), T.reshape(4, 4), err.value, inl.value
