"""Python host-side mirror of the reference's `dso::TrackerAndScaler` public surface
(src/scale_optimization/TrackerAndScaler.h:38-64) on top of the C ABI.  Used by tests/ and
bench.py; the C++ adaptor for the ROS node is direct_stereo_slam_amd/host/TrackerAndScaler.hpp.

Method names follow the reference: makeK, setCoarseTrackingRef (takes the template lists that
makeCoarseDepthL0 produced), scaleCoarseDepthL0, trackNewestCoarse, optimizeScale; the public
output fields refFrameID / lastFlowIndicators keep their names.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MAX_LEVELS, Params, Stats, c_double_p, c_float_p, c_int_p, check


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


def pinned_array(shape, dtype=np.float32):
    """numpy array backed by pinned host memory (dsm_host_alloc): dsm_tracker_upload_image copies from it by DMA.
    The memory is released when the array is garbage collected."""
    L = _lib.load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    check(L.dsm_host_alloc(n, C.byref(p)))
    buf = (C.c_char * n).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                L.dsm_host_free(self.ptr)
            except Exception:
                pass

    owner = _Owner(p)
    out = arr.view()
    _PINNED_OWNERS[id(out)] = (owner, buf)
    import weakref

    weakref.finalize(out, _PINNED_OWNERS.pop, id(out), None)
    return out


_PINNED_OWNERS = {}


def default_params():
    p = Params()
    check(_lib.load().dsm_params_default_sized(C.byref(p), C.sizeof(Params)))  # (a layout drift of this mirror fails here)
    return p


class Context:
    """device + stream + batch workspaces (dsm_context)"""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        check(self.L.dsm_context_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dsm_context_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def sync(self):
        check(self.L.dsm_context_sync(self.h))

    def set_timing(self, on):
        check(self.L.dsm_context_set_timing(self.h, int(on)))

    def set_streams(self, n):
        check(self.L.dsm_context_set_streams(self.h, int(n)))

    def stream_queues(self):
        """(group streams in use, how many of them share a hardware queue with another group): dsm_context_stream_queues"""
        a, b = C.c_int(), C.c_int()
        check(self.L.dsm_context_stream_queues(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self):
        s = Stats()
        check(self.L.dsm_context_get_stats(self.h, C.byref(s)))
        return s

    def stats2(self):
        """statistics of the scale segment of the last track_and_scale_batch"""
        s = Stats()
        check(self.L.dsm_context_get_stats2(self.h, C.byref(s)))
        return s

    def read_bandwidth(self, nbytes=1 << 30, iters=10):
        """measurement aid: read-only streaming bandwidth in GB/s"""
        g = C.c_double()
        check(self.L.dsm_diag_read_bandwidth(self.h, nbytes, iters, C.byref(g)))
        return g.value

    def xwg_litmus(self, pairs=128, iters=80000):
        """(hand-offs, stale words) of the cross-workgroup hand-off litmus (dsm_diag_xwg_litmus)"""
        n, bad = C.c_longlong(), C.c_longlong()
        check(self.L.dsm_diag_xwg_litmus(self.h, pairs, iters, C.byref(n), C.byref(bad)))
        return n.value, bad.value

    def read_bandwidth_chunked(self, nbytes=1 << 30, chunk_bytes=112 * 1024, iters=10):
        """measurement aid: same, one contiguous chunk per workgroup (the eval kernels' access pattern)"""
        g = C.c_double()
        check(self.L.dsm_diag_read_bandwidth_chunked(self.h, nbytes, chunk_bytes, iters, C.byref(g)))
        return g.value

    # ---- batched forms -----------------------------------------------------------------
    def track_batch(self, trackers, poses, affs, coarsest, min_res=None):
        n = len(trackers)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        poses = np.ascontiguousarray(poses, np.float64).reshape(n, 7).copy()
        affs = np.ascontiguousarray(affs, np.float64).reshape(n, 2).copy()
        mr = None
        if min_res is not None:
            mr = np.ascontiguousarray(min_res, np.float64).reshape(n, MAX_LEVELS)
        last = np.zeros((n, MAX_LEVELS))
        flow = np.zeros((n, 3))
        good = np.zeros(n, np.int32)
        check(self.L.dsm_track_batch(self.h, n, hs, _dp(poses), _dp(affs), coarsest, None if mr is None else _dp(mr),
                                     _dp(last), _dp(flow), good.ctypes.data_as(c_int_p)))
        return good.astype(bool), poses, affs, last, flow

    def upload_images(self, trackers, slots, images, exposures=None, asynchronous=False):
        """dsm_upload_images[_async]: hand over one level-0 image (float32 or uint8, all of one type) per (tracker, slot)
        in one call.  Images may be row-strided views (e.g. the calibration crop of a larger camera image), all with one
        pitch.  Slots 2 / 3 (DSM_SLOT_NEXT_LEFT / RIGHT) are the back buffers that advance_frames swaps in.
        asynchronous=True returns at once; the images must not be modified until upload_wait()."""
        n = len(trackers)
        if n == 0:
            return
        dt = np.dtype(images[0].dtype)
        if dt not in (np.dtype(np.float32), np.dtype(np.uint8)):
            raise TypeError("upload_images: float32 or uint8 images")
        keep, pitch = [], None
        for im, t in zip(images, trackers):
            if im.dtype != dt or im.shape != (t.hgt, t.w):
                raise ValueError("upload_images: image type / shape mismatch")
            if im.strides[1] != dt.itemsize or im.strides[0] < t.w * dt.itemsize:
                im = np.ascontiguousarray(im)
            if pitch is None:
                pitch = im.strides[0]
            elif im.strides[0] != pitch:
                raise ValueError("upload_images: images of one call share one row pitch")
            keep.append(im)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        ps = (C.c_void_p * n)(*[im.ctypes.data for im in keep])
        sl = np.ascontiguousarray(slots, np.int32)
        ex = np.ones(n, np.float32) if exposures is None else np.ascontiguousarray(exposures, np.float32)
        fn = self.L.dsm_upload_images_async if asynchronous else self.L.dsm_upload_images
        check(fn(self.h, n, hs, sl.ctypes.data_as(c_int_p), ps, _fp(ex), 1 if dt == np.dtype(np.uint8) else 0, pitch))
        if asynchronous:
            self._pending_images = keep  # alive until upload_wait

    def upload_wait(self):
        check(self.L.dsm_upload_wait(self.h))
        self._pending_images = None

    def advance_frames(self, trackers, slots):
        """dsm_frames_advance: swap the back buffers of the given (tracker, slot in {0, 1}) pairs in"""
        n = len(trackers)
        if n == 0:
            return
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        sl = np.ascontiguousarray(slots, np.int32)
        check(self.L.dsm_frames_advance(self.h, n, hs, sl.ctypes.data_as(c_int_p)))

    def track_and_scale_batch(self, trackers, poses, affs, coarsest, scale_trackers, scales, min_res=None):
        """dsm_track_and_scale_batch: the frames' tracking and the keyframes' scale optimisation in one call (the scale
        problems run on their own stream under the tracking kernels); returns track_batch's tuple + (err, scales)"""
        n, n2 = len(trackers), len(scale_trackers)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        hs2 = (C.c_void_p * max(1, n2))(*[t.h for t in scale_trackers])
        poses = np.ascontiguousarray(poses, np.float64).reshape(n, 7).copy()
        affs = np.ascontiguousarray(affs, np.float64).reshape(n, 2).copy()
        mr = None if min_res is None else np.ascontiguousarray(min_res, np.float64).reshape(n, MAX_LEVELS)
        last, flow, good = np.zeros((n, MAX_LEVELS)), np.zeros((n, 3)), np.zeros(n, np.int32)
        sc = np.ascontiguousarray(scales, np.float32).reshape(n2).copy()
        err = np.zeros(max(1, n2), np.float32)
        check(self.L.dsm_track_and_scale_batch(self.h, n, hs, _dp(poses), _dp(affs), coarsest, None if mr is None else _dp(mr), _dp(last),
                                               _dp(flow), good.ctypes.data_as(c_int_p), n2, hs2, _fp(sc) if n2 else None, _fp(err)))
        return good.astype(bool), poses, affs, last, flow, err[:n2], sc

    def optimize_scale_batch(self, trackers, scales, coarsest):
        n = len(trackers)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        sc = np.ascontiguousarray(scales, np.float32).reshape(n).copy()
        err = np.zeros(n, np.float32)
        check(self.L.dsm_optimize_scale_batch(self.h, n, hs, _fp(sc), coarsest, _fp(err)))
        return err, sc


class Stream:
    """Streaming form of Context.track_and_scale_batch (dsm_stream_*): a pool of resident problems advanced in passes; problems
    are admitted as slots free up and retire individually.  Results are bit-identical to the batch calls."""

    def __init__(self, ctx, track_slots, scale_slots=0, engine=None, ticks=0):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        check(self.L.dsm_stream_create(ctx.h, int(track_slots), int(scale_slots), C.byref(h)))
        self.h = h
        if engine is not None:
            self.set_engine(engine, ticks)

    def set_engine(self, engine, ticks=0):
        """0: passes with carried stragglers; 1 (the library's default): ticks (one LM round per resident problem and tick, device-side admission).
        ticks = 0 keeps what is in force (initially: every advance sized by the stream from the retired problems' mean life)"""
        check(self.L.dsm_stream_set_engine(self.h, int(engine), int(ticks)))

    def set_chain(self, max_rounds):
        """tick engine: LM rounds a problem whose pending evaluation is ONE chunk may run inside one tick, in the workgroup that evaluates
        it (dsm_stream_set_chain); 0 = off, -1 = the library's default.  Scheduling only."""
        check(self.L.dsm_stream_set_chain(self.h, int(max_rounds)))

    def close(self):
        if getattr(self, "h", None):
            self.L.dsm_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def submit_track(self, trackers, poses, affs, coarsest, min_res=None):
        n = len(trackers)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        poses = np.ascontiguousarray(poses, np.float64).reshape(n, 7)
        affs = np.ascontiguousarray(affs, np.float64).reshape(n, 2)
        mr = None if min_res is None else np.ascontiguousarray(min_res, np.float64).reshape(n, MAX_LEVELS)
        tk = (C.c_uint64 * n)()
        check(self.L.dsm_stream_submit_track(self.h, n, hs, _dp(poses), _dp(affs), int(coarsest), None if mr is None else _dp(mr), tk))
        return list(tk)

    def submit_scale(self, trackers, scales, coarsest):
        n = len(trackers)
        hs = (C.c_void_p * n)(*[t.h for t in trackers])
        sc = np.ascontiguousarray(scales, np.float32).reshape(n)
        tk = (C.c_uint64 * n)()
        check(self.L.dsm_stream_submit_scale(self.h, n, hs, _fp(sc), int(coarsest), tk))
        return list(tk)

    def advance(self):
        check(self.L.dsm_stream_advance(self.h))

    def drain(self):
        check(self.L.dsm_stream_drain(self.h))

    def sync(self):
        """wait for everything in flight and read it back (the tick engine's advances are pipelined)"""
        check(self.L.dsm_stream_sync(self.h))

    def set_pipelined(self, on):
        check(self.L.dsm_stream_set_pipelined(self.h, int(bool(on))))

    def counts(self):
        """(resident, waiting, results ready)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self.L.dsm_stream_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def results(self, max_results=None):
        """retired problems, oldest first: list of _lib.StreamResult"""
        if max_results is None:
            max_results = self.counts()[2]
        if max_results <= 0:
            return []
        buf = (_lib.StreamResult * max_results)()
        n = C.c_int()
        check(self.L.dsm_stream_results(self.h, max_results, buf, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def set_quantile(self, q, lvl=-1):
        """q: one value (level lvl, or every level when lvl < 0) or a list from level 0"""
        if np.ndim(q) == 0:
            check(self.L.dsm_stream_set_quantile(self.h, int(lvl), float(q)))
        else:
            for l, v in enumerate(q):
                check(self.L.dsm_stream_set_quantile(self.h, l, float(v)))

    def set_rounds(self, mode, rounds):
        arr = None if rounds is None else (C.c_int * MAX_LEVELS)(*(list(rounds) + [0] * MAX_LEVELS)[:MAX_LEVELS])
        check(self.L.dsm_stream_set_rounds(self.h, int(mode), arr))

    def stats(self):
        a, b = Stats(), Stats()
        check(self.L.dsm_stream_get_stats(self.h, C.byref(a), C.byref(b)))
        return a, b

    def schedule(self, mode=0):
        r = (C.c_int * MAX_LEVELS)()
        p, q, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        check(self.L.dsm_stream_get_schedule(self.h, int(mode), r, C.byref(p), C.byref(q), C.byref(c)))
        return dict(rounds=list(r), passes=p.value, retired=q.value, carried_slot_passes=c.value)


class TrackerAndScaler:
    def __init__(self, ctx, w, h, nlevels, tfm_vec, K1, params=None):
        """tfm_vec: 16 doubles of T_stereo (cams/*/T_stereo.yaml); K1 = (fx,fy,cx,cy) of camera 1
        (reference ctor, TrackerAndScaler.cpp:47-109)."""
        self.ctx = ctx
        self.L = ctx.L
        self.w, self.hgt, self.nlevels = w, h, nlevels
        self.params = params if params is not None else default_params()
        T = np.ascontiguousarray(np.asarray(tfm_vec, np.float64).reshape(16))
        K1 = np.ascontiguousarray(np.asarray(K1, np.float32))
        hnd = C.c_void_p()
        check(self.L.dsm_tracker_create(ctx.h, w, h, nlevels, _dp(T), _fp(K1), C.byref(self.params), C.byref(hnd)))
        self.h = hnd
        self.lastFlowIndicators = np.full(3, 1000.0)
        self.firstCoarseRMSE = -1.0

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.L.dsm_tracker_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    @property
    def refFrameID(self):
        return self.L.dsm_tracker_ref_frame_id(self.h)

    def makeK(self, fx, fy, cx, cy):
        check(self.L.dsm_tracker_make_k(self.h, fx, fy, cx, cy))

    def setCoarseTrackingRef(self, ref_frame_id, ref_aff, ref_exposure, pc_u, pc_v, pc_idepth, pc_color):
        n = (C.c_int * self.nlevels)(*[len(a) for a in pc_u])
        arrs = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (pc_u, pc_v, pc_idepth, pc_color)]
        check(self.L.dsm_tracker_set_ref(self.h, ref_frame_id, float(ref_aff[0]), float(ref_aff[1]), ref_exposure, n,
                                         *[_ptr_array(a) for a in arrs]))
        self.firstCoarseRMSE = -1.0

    def setCoarseTrackingRefFromPoints(self, ref_frame_id, ref_aff, ref_exposure, pu, pv, pidepth, pweight,
                                       frame_owner=None, slot=0):
        """makeCoarseDepthL0 + setCoarseTrackingRef (TrackerAndScaler.cpp:143-327) on the device (row N3): the
        window's active points as flat arrays; the keyframe's pyramid is the one resident in `frame_owner`'s
        `slot` (default: this tracker's new-left slot).  Returns pc_n per level."""
        arrs = [np.ascontiguousarray(a, np.float32) for a in (pu, pv, pidepth, pweight)]
        npts = len(arrs[0])
        assert all(len(a) == npts for a in arrs)
        n = (C.c_int * self.nlevels)()
        owner = self if frame_owner is None else frame_owner
        check(self.L.dsm_tracker_set_ref_from_points(self.h, owner.h, slot, ref_frame_id, float(ref_aff[0]), float(ref_aff[1]),
                                                     ref_exposure, npts, *[_fp(a) for a in arrs], n))
        self.firstCoarseRMSE = -1.0
        return list(n)

    @staticmethod
    def setCoarseTrackingRefsFromPoints(ctx, jobs):
        """dsm_set_refs_from_points: the device templates of several trackers' new keyframes in ONE call (one host synchronisation).
        jobs: dicts with tracker, ref_frame_id, ref_aff, ref_exposure, pu, pv, pidepth, pweight and optionally frame_owner, slot.
        Returns pc_n per level for every job."""
        from ._lib import RefJob

        arr = (RefJob * len(jobs))()
        keep, outs = [], []
        for J, j in zip(arr, jobs):
            t = j["tracker"]
            a = [np.ascontiguousarray(j[k], np.float32) for k in ("pu", "pv", "pidepth", "pweight")]
            n = (C.c_int * t.nlevels)()
            keep.append(a)
            outs.append(n)
            J.t, J.frame_owner = t.h, j.get("frame_owner", t).h
            J.slot, J.ref_frame_id = j.get("slot", 0), j["ref_frame_id"]
            J.ref_aff_a, J.ref_aff_b, J.ref_exposure, J.npts = float(j["ref_aff"][0]), float(j["ref_aff"][1]), j["ref_exposure"], len(a[0])
            J.pu, J.pv, J.pidepth, J.pweight = [_fp(x) for x in a]
            J.n_out = C.cast(n, C.POINTER(C.c_int))
        check(ctx.L.dsm_set_refs_from_points(ctx.h, len(jobs), arr))
        for j in jobs:
            j["tracker"].firstCoarseRMSE = -1.0
        return [list(n) for n in outs]

    def scaleCoarseDepthL0(self, scale):
        check(self.L.dsm_tracker_scale_depth(self.h, scale))

    def get_template(self, lvl):
        n = C.c_int()
        check(self.L.dsm_tracker_get_template(self.h, lvl, C.byref(n), None, None, None, None))
        out = [np.zeros(n.value, np.float32) for _ in range(4)]
        check(self.L.dsm_tracker_get_template(self.h, lvl, C.byref(n), *[_fp(a) for a in out]))
        return out

    def upload_frame(self, slot, dIp, ab_exposure=1.0):
        dIp = [np.ascontiguousarray(a, np.float32) for a in dIp]
        check(self.L.dsm_tracker_upload_frame(self.h, slot, _ptr_array(dIp), ab_exposure))

    def upload_intensity(self, slot, planes, ab_exposure=1.0):
        """the intensity channel of every level alone (dsm_tracker_upload_intensity)"""
        planes = [np.ascontiguousarray(a, np.float32) for a in planes]
        check(self.L.dsm_tracker_upload_intensity(self.h, slot, _ptr_array(planes), ab_exposure))

    def upload_image(self, slot, image, ab_exposure=1.0):
        image = np.ascontiguousarray(image, np.float32)
        assert image.shape == (self.hgt, self.w)
        check(self.L.dsm_tracker_upload_image(self.h, slot, _fp(image), ab_exposure))

    def get_frame(self, slot, lvl):
        out = np.zeros((self.hgt >> lvl, self.w >> lvl, 3), np.float32)
        check(self.L.dsm_tracker_get_frame(self.h, slot, lvl, _fp(out)))
        return out

    def calcResPose(self, lvl, pose, aff, cutoff):
        """fused calcResPose + calcGSSSEPose: returns (rs[6], H[8,8], b[8], n_warped)"""
        pose = np.ascontiguousarray(pose, np.float64)
        aff = np.ascontiguousarray(aff, np.float64)
        rs, H, b, n = np.zeros(6), np.zeros(64), np.zeros(8), C.c_int()
        check(self.L.dsm_tracker_calc_res_pose(self.h, lvl, _dp(pose), _dp(aff), cutoff, _dp(rs), _dp(H), _dp(b), C.byref(n)))
        return rs, H.reshape(8, 8), b, n.value

    def calcResScale(self, lvl, scale, cutoff):
        rs, H, b, n = np.zeros(6), C.c_float(), C.c_float(), C.c_int()
        check(self.L.dsm_tracker_calc_res_scale(self.h, lvl, scale, cutoff, _dp(rs), C.byref(H), C.byref(b), C.byref(n)))
        return rs, H.value, b.value, n.value

    def trackNewestCoarse(self, lastToNew, aff_g2l, coarsestLvl, minResForAbort=None):
        """returns (good, lastToNew_out, aff_g2l_out, lastResiduals) and sets lastFlowIndicators"""
        pose = np.array(lastToNew, np.float64)
        aff = np.array(aff_g2l, np.float64)
        mr = None if minResForAbort is None else np.ascontiguousarray(minResForAbort, np.float64)
        last, flow, good = np.zeros(MAX_LEVELS), np.zeros(3), C.c_int()
        check(self.L.dsm_tracker_track(self.h, _dp(pose), _dp(aff), coarsestLvl, None if mr is None else _dp(mr), _dp(last),
                                       _dp(flow), C.byref(good)))
        self.lastFlowIndicators = flow
        return bool(good.value), pose, aff, last

    def optimizeScale(self, scale, coarsestLvl):
        """returns (error, scale_out) as the reference's `float optimizeScale(fh1, float &scale, lvl)`"""
        s, err = C.c_float(scale), C.c_float()
        check(self.L.dsm_tracker_optimize_scale(self.h, C.byref(s), coarsestLvl, C.byref(err)))
        return err.value, s.value

    def optimizeScaleGuesses(self, guesses, coarsestLvl):
        """the untrapped branch of FrontEnd::optimizeScale (FrontEnd.cpp:995-1003) as one batched call: returns
        (scale_error, new_scale, all_errors, all_scales)"""
        g = np.ascontiguousarray(guesses, np.float32)
        s, e = C.c_float(), C.c_float()
        sa, ea = np.zeros(len(g), np.float32), np.zeros(len(g), np.float32)
        check(self.L.dsm_tracker_optimize_scale_guesses(self.h, len(g), _fp(g), coarsestLvl, C.byref(s), C.byref(e), _fp(sa), _fp(ea)))
        return e.value, s.value, ea, sa

    def reduction_geometry(self, lvl, n):
        t, p, c = C.c_int(), C.c_int(), C.c_int()
        check(self.L.dsm_reduction_geometry(self.h, lvl, n, C.byref(t), C.byref(p), C.byref(c)))
        return t.value, p.value, c.value


def make_coarse_depth_l0(w, h, nlevels, pu, pv, pidepth, pweight, ref_dIp):
    """TrackerAndScaler::makeCoarseDepthL0 (TrackerAndScaler.cpp:143-315) from flat active-point arrays,
    through the C ABI (host code in this round).  Returns [pc_u, pc_v, pc_idepth, pc_color] lists per level,
    i.e. the arguments of setCoarseTrackingRef."""
    L = _lib.load()
    pu, pv, pidepth, pweight = [np.ascontiguousarray(a, np.float32) for a in (pu, pv, pidepth, pweight)]
    ref = [np.ascontiguousarray(a, np.float32) for a in ref_dIp]
    n_out = (C.c_int * nlevels)()
    outs = [[np.zeros((w >> l) * (h >> l), np.float32) for l in range(nlevels)] for _ in range(4)]
    check(L.dsm_make_coarse_depth_l0(w, h, nlevels, len(pu), _fp(pu), _fp(pv), _fp(pidepth), _fp(pweight), _ptr_array(ref),
                                     n_out, *[_ptr_array(o) for o in outs]))
    return [[o[l][: n_out[l]].copy() for l in range(nlevels)] for o in outs]


class PoseEstimator:
    """Python mirror of the reference's `dso::PoseEstimator` (PoseEstimator.h:34-83) on the C ABI."""

    def __init__(self, ctx, w, h, nlevels, params=None):
        self.ctx, self.L = ctx, ctx.L
        self.params = params if params is not None else default_params()
        hnd = C.c_void_p()
        check(self.L.dsm_pose_estimator_create(ctx.h, w, h, nlevels, C.byref(self.params), C.byref(hnd)))
        self.h = hnd

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.L.dsm_pose_estimator_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def estimate(self, pts_xyz, ref_colors, ref_ab_exposure, new_dIp, new_ab_exposure, new_cam, coarsest_lvl, ref_to_new):
        """returns (ok, ref_to_new 4x4, pose_error) like `bool estimate(pts, ref_ab_exposure, new_fh, new_cam,
        coarsest_lvl, Matrix4d& ref_to_new, float& pose_error)`"""
        xyz = np.ascontiguousarray(pts_xyz, np.float64).reshape(-1, 3)
        cols = [np.ascontiguousarray(c, np.float32) for c in ref_colors]
        dIp = [np.ascontiguousarray(a, np.float32) for a in new_dIp]
        cam = np.ascontiguousarray(new_cam, np.float32)
        T = np.ascontiguousarray(ref_to_new, np.float64).reshape(16).copy()
        err, ok = C.c_float(), C.c_int()
        check(self.L.dsm_pose_estimator_estimate(self.h, len(xyz), _dp(xyz), _ptr_array(cols), ref_ab_exposure, _ptr_array(dIp),
                                                 new_ab_exposure, _fp(cam), coarsest_lvl, _dp(T), C.byref(err), C.byref(ok)))
        return bool(ok.value), T.reshape(4, 4), err.value


def track_hypotheses(ctx, trk, tries, aff_last_2_l, coarsestLvl, last_coarse_rmse0, reTrackThreshold=1.5):
    """The hypothesis loop of FrontEnd::trackNewCoarse (FrontEnd.cpp:194-247) with the same results as
    the reference's sequential loop, but evaluated as ONE batched launch sequence ("next" row N4).

    Reference semantics: try i is tracked with minResForAbort = achievedRes of the earlier tries and is
    aborted at the first level whose residual exceeds 1.5x that value (TrackerAndScaler.cpp:598); the
    loop ends at the first success below last_coarse_rmse*reTrackThreshold.  Here try 0 runs alone (the
    common case ends there); otherwise all remaining tries run as one batch WITHOUT abort, and the abort
    / take-over logic is replayed on the host from their per-level residuals -- a level's LM result does
    not depend on minResForAbort, so the replay is exact.
    Returns (haveOneGood, lastF_2_fh, aff_g2l, flowVecs, achievedRes, tries_used)."""
    tries = np.ascontiguousarray(tries, np.float64).reshape(-1, 7)
    n = len(tries)
    achieved = np.full(MAX_LEVELS, np.nan)
    have_good = False
    flow = np.array([100.0, 100.0, 100.0])
    best_pose, best_aff = IDENTITY_POSE7.copy(), np.zeros(2)

    def consume(good, pose, aff, cur_res, fl):
        nonlocal have_good, flow, best_pose, best_aff
        if good and np.isfinite(np.float32(cur_res[0])) and not (cur_res[0] >= achieved[0]):  # :225-233
            flow, best_aff, best_pose, have_good = fl.copy(), aff.copy(), pose.copy(), True
        if have_good:  # :236-243
            for l in range(5):
                if not np.isfinite(np.float32(achieved[l])) or achieved[l] > cur_res[l]:
                    achieved[l] = cur_res[l]
        return have_good and achieved[0] < last_coarse_rmse0 * reTrackThreshold  # :245-247

    # try 0 alone
    good, pose, aff, last = trk.trackNewestCoarse(tries[0], aff_last_2_l, coarsestLvl, achieved)
    used = 1
    if consume(good, pose, aff, last, trk.lastFlowIndicators) or n == 1:
        return have_good, (best_pose if have_good else tries[0]), (best_aff if have_good else np.asarray(aff_last_2_l, float)), \
            (flow if have_good else np.zeros(3)), achieved, used
    # the rest as one batch, no abort
    m = n - 1
    goods, poses, affs, lasts, flows = ctx.track_batch([trk] * m, tries[1:], np.tile(np.asarray(aff_last_2_l, np.float64), (m, 1)),
                                                       coarsestLvl, None)
    for i in range(m):
        used += 1
        cur = lasts[i].copy()
        good_i = bool(goods[i])
        # replay the abort test of TrackerAndScaler.cpp:598 level by level (coarse to fine)
        for l in range(coarsestLvl, -1, -1):
            if cur[l] > 1.5 * achieved[l]:
                cur[:l] = np.nan
                good_i = False
                break
        pose_i = poses[i] if good_i else tries[1 + i]  # an aborted try leaves lastToNew_out untouched
        if consume(good_i, pose_i, affs[i], cur, flows[i]):
            break
    if not have_good:  # :249-256
        return False, tries[0], np.asarray(aff_last_2_l, float), np.zeros(3), achieved, used
    return True, best_pose, best_aff, flow, achieved, used


IDENTITY_POSE7 = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
