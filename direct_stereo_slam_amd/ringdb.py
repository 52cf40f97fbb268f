"""Ring-key database: host mirror of `search_ringkey` (src/loop_closure/loop_detection/search_place.h:25-57)
on the C ABI, plus the cross-GPU merge for the sharded DB.

The device scan lives in csrc/ringkey_kernels.hip, the cross-shard merge -- the only collective on the whole hot
path -- in csrc/comm_capi.hip behind the C ABI: `Comm` wraps dsm_comm (RCCL, loaded by the library itself) and
`RingKeyDB.merge_topk_device` calls dsm_ringdb_merge_topk: k rounds of all-reduce(min) over packed
(dist2 << 32 | global index) candidates with winner pop, or one all-gather + local merge.  A slot-wise min of sorted
triples would NOT be a correct top-k.  (The torch restatement of the round algorithm that the gloo plumbing test uses on hosts
without a GPU lives with the tests: tests/_merge_ref.py.)  `RingKeyDB.merge_topk_with` runs the C ABI's merge kernels over a caller-supplied
transport (threads or gloo processes sharing one GPU in the tests).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ALLGATHER_FN, ALLREDUCE_MIN_FN, MERGE_ALGOS, NO_CANDIDATE, c_float_p, c_int64_p, c_int_p, check


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def unpack(packed):
    """packed int64 -> (dist2 float32, global index int64); NO_CANDIDATE -> (inf, -1)"""
    packed = np.asarray(packed, np.int64)
    idx = (packed & 0xFFFFFFFF).astype(np.int64)
    bits = (packed >> 32).astype(np.uint32)
    dist = bits.view(np.float32).astype(np.float32)
    none = packed == NO_CANDIDATE
    return np.where(none, np.float32(np.inf), dist), np.where(none, -1, idx)


def candidates_from_packed(packed_row):
    """threshold already applied on the device; drop the dummy (idx > 0) and emit idx-1
    (search_place.h:34-38)"""
    out = []
    for p in np.asarray(packed_row, np.int64):
        if p == NO_CANDIDATE:
            continue
        idx = int(p & 0xFFFFFFFF)
        if idx > 0:
            out.append(idx - 1)
    return out


class Comm:
    """dsm_comm: one rank per GPU over RCCL.  rank 0 calls Comm.unique_id() and hands the 128 bytes to every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        check(_lib.load().dsm_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, unique_id, rank, nranks):
        self.ctx, self.L = ctx, ctx.L
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(self.L.dsm_comm_create(ctx.h, buf, rank, nranks, C.byref(h)))
        self.h, self.rank, self.nranks = h, rank, nranks

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.L.dsm_comm_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()


class RingKeyDB:
    """flann::Index replacement + delay queue.  shard_rank/shard_count: this handle stores only the
    global ordinals with ordinal % shard_count == shard_rank."""

    def __init__(self, ctx, dim=20, margin=100, k=3, thres=0.1, dummy=None, capacity=1024, shard_rank=0, shard_count=1):
        self.ctx, self.L = ctx, ctx.L
        self.dim, self.k = dim, k
        self.shard_rank, self.shard_count = shard_rank, shard_count
        d = None if dummy is None else _fp(np.ascontiguousarray(dummy, np.float32))
        h = C.c_void_p()
        check(self.L.dsm_ringdb_create(ctx.h, dim, margin, k, thres, d, capacity, shard_rank, shard_count, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.L.dsm_ringdb_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def size(self):
        return self.L.dsm_ringdb_size(self.h)

    def add_points(self, keys):
        keys = np.ascontiguousarray(keys, np.float32).reshape(-1, self.dim)
        check(self.L.dsm_ringdb_add_points(self.h, _fp(keys), keys.shape[0]))

    def enqueue(self, key):
        key = np.ascontiguousarray(key, np.float32)
        check(self.L.dsm_ringdb_enqueue(self.h, _fp(key)))

    def attach_comm(self, comm):
        """sharded handle: search_ringkey becomes a collective over `comm` (None detaches)"""
        check(self.L.dsm_ringdb_attach_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm

    def attach_transport(self, nranks, allreduce_min):
        """sharded handle: search_ringkey becomes a collective over a caller-supplied all-reduce(min)
        (allreduce_min(d_buf_ptr, count, stream_ptr) on a device buffer of unsigned 64-bit words); None detaches"""
        if allreduce_min is None:
            check(self.L.dsm_ringdb_attach_transport(self.h, 0, C.cast(None, ALLREDUCE_MIN_FN), C.cast(None, ALLGATHER_FN), None))
            self._tr = None
            return

        def _ar(user, buf, count, stream):
            try:
                allreduce_min(buf, count, stream)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return -1

        self._tr = ALLREDUCE_MIN_FN(_ar)  # keep the callback alive as long as it is attached
        check(self.L.dsm_ringdb_attach_transport(self.h, nranks, self._tr, C.cast(None, ALLGATHER_FN), None))

    def merge_topk_device(self, comm, d_packed_ptr, nq, algo="allreduce_min"):
        """cross-shard merge of the nq x k packed candidates at the raw device pointer, in place (dsm_ringdb_merge_topk);
        asynchronous on the context stream"""
        check(self.L.dsm_ringdb_merge_topk(self.h, comm.h, C.c_void_p(d_packed_ptr), nq, MERGE_ALGOS[algo]))

    def merge_topk_with(self, d_packed_ptr, nq, nranks, allreduce_min=None, allgather=None, algo="allreduce_min"):
        """the same merge kernels over a caller-supplied transport: allreduce_min(d_buf_ptr, count, stream_ptr) /
        allgather(d_send_ptr, d_recv_ptr, count, stream_ptr) act on device buffers of unsigned 64-bit words"""
        def _ar(user, buf, count, stream):
            try:
                allreduce_min(buf, count, stream)
                return 0
            except Exception:  # noqa: BLE001 -- reported through the C ABI's return code
                import traceback

                traceback.print_exc()
                return -1

        def _ag(user, send, recv, count, stream):
            try:
                allgather(send, recv, count, stream)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return -1

        far = ALLREDUCE_MIN_FN(_ar) if allreduce_min else C.cast(None, ALLREDUCE_MIN_FN)
        fag = ALLGATHER_FN(_ag) if allgather else C.cast(None, ALLGATHER_FN)
        check(self.L.dsm_ringdb_merge_topk_with(self.h, C.c_void_p(d_packed_ptr), nq, MERGE_ALGOS[algo], nranks, far, fag, None))

    def search_ringkey(self, key):
        """search_place.h:25-57: returns the candidate list (a collective call on a sharded handle with a communicator)"""
        key = np.ascontiguousarray(key, np.float32)
        cand = (C.c_int * self.k)()
        nc = C.c_int()
        check(self.L.dsm_ringdb_query_then_enqueue(self.h, _fp(key), cand, C.byref(nc)))
        return [cand[i] for i in range(nc.value)]

    def knn_packed_host(self, queries):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        out = np.zeros((q.shape[0], self.k), np.int64)
        check(self.L.dsm_ringdb_knn_packed_host(self.h, _fp(q), q.shape[0], out.ctypes.data_as(c_int64_p)))
        return out

    def knn_packed_device(self, d_queries_ptr, nq, d_out_ptr):
        """queries / output are raw device pointers (e.g. torch tensors' data_ptr()); asynchronous on
        the context stream -- call ctx.sync() before another stream consumes the result."""
        check(self.L.dsm_ringdb_knn_packed_dev(self.h, C.c_void_p(d_queries_ptr), nq, C.c_void_p(d_out_ptr)))


def scancontext_generate(pts_spherical, lidar_range, num_s=60, num_r=20):
    """ScanContext::generate (ScanContext.cpp:78-141) through the C ABI (host code by design).
    Returns (ringkey[num_r] float32, sig_idx int32, sig_val float64, tfm_pca_rig 4x4)."""
    L = _lib.load()
    pts = np.ascontiguousarray(pts_spherical, np.float64).reshape(-1, 3)
    ringkey = np.zeros(num_r, np.float32)
    sig_idx = np.zeros(num_s * num_r, np.int32)
    sig_val = np.zeros(num_s * num_r, np.float64)
    n_sig = C.c_int()
    tfm = np.zeros(16)
    check(L.dsm_scancontext_generate(pts.ctypes.data_as(_lib.c_double_p), len(pts), lidar_range, num_s, num_r, _fp(ringkey),
                                     sig_idx.ctypes.data_as(c_int_p), sig_val.ctypes.data_as(_lib.c_double_p), C.byref(n_sig),
                                     tfm.ctypes.data_as(_lib.c_double_p)))
    return ringkey, sig_idx[: n_sig.value].copy(), sig_val[: n_sig.value].copy(), tfm.reshape(4, 4)


def generate_spherical_points(kf_ids, kf_pose_wc, cur_cw, lidar_range, pt_kf_id, pt_xyz):
    """generate_spherical_points (generate_spherical_points.h:27-85) through the C ABI (host code).
    Returns (kf_keep[n_kf] bool, sel_idx int32 -- the reference's updated pts_nearby -- and pts_spherical (n,3))."""
    L = _lib.load()
    kf_ids = np.ascontiguousarray(kf_ids, np.int32)
    poses = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 6)
    cw = np.ascontiguousarray(cur_cw, np.float64).reshape(3, 4)
    pid = np.ascontiguousarray(pt_kf_id, np.int32)
    xyz = np.ascontiguousarray(pt_xyz, np.float64).reshape(-1, 3)
    keep = np.zeros(len(kf_ids), np.int32)
    sel = np.zeros(max(1, len(pid)), np.int32)
    out = np.zeros((max(1, len(pid)), 3))
    n = C.c_int()
    check(L.dsm_generate_spherical_points(len(kf_ids), kf_ids.ctypes.data_as(c_int_p), poses.ctypes.data_as(_lib.c_double_p),
                                          cw.ctypes.data_as(_lib.c_double_p), lidar_range, len(pid), pid.ctypes.data_as(c_int_p),
                                          xyz.ctypes.data_as(_lib.c_double_p), keep.ctypes.data_as(c_int_p), C.byref(n),
                                          sel.ctypes.data_as(c_int_p), out.ctypes.data_as(_lib.c_double_p)))
    return keep.astype(bool), sel[: n.value].copy(), out[: n.value].copy()


class LoopBatch:
    """The ctypes job table of dsm_loop_descriptors_batch / dsm_loop_detect_batch and its output arrays, built once: `run()` is the C call
    alone (a node keeps its clouds and job tables; bench.py times this), `results()` unpacks."""

    def __init__(self, ctx, jobs, lidar_range, num_s=60, num_r=20, scancontext=True, db=None, selected_points=True, pinned_clouds=False):
        """pinned_clouds: the clouds (pt_kf_id, pt_xyz) are kept in page-locked memory (dsm_host_alloc), which the device reads directly:
        no staging copy on the host"""
        self.ctx, self.L, self.db = ctx, ctx.L, db
        self.lidar_range, self.num_s, self.num_r, self.scancontext, self.selected_points = lidar_range, num_s, num_r, scancontext, selected_points
        self.arr = (_lib.LoopJob * len(jobs))()
        self.keepalive, self.outs = [], []
        dp, ip = _lib.c_double_p, c_int_p
        for j, (kf_ids, kf_pose_wc, cur_cw, pt_kf_id, pt_xyz) in enumerate(jobs):
            kf_ids = np.ascontiguousarray(kf_ids, np.int32)
            poses = np.ascontiguousarray(kf_pose_wc, np.float64).reshape(-1, 6)
            cw = np.ascontiguousarray(cur_cw, np.float64).reshape(3, 4)
            pid = np.ascontiguousarray(pt_kf_id, np.int32)
            xyz = np.ascontiguousarray(pt_xyz, np.float64).reshape(-1, 3)
            if (pinned_clouds[j] if isinstance(pinned_clouds, (list, tuple)) else pinned_clouds) and len(pid):
                from .tracker import pinned_array

                ppid, pxyz = pinned_array(pid.shape, np.int32), pinned_array(xyz.shape, np.float64)
                ppid[...], pxyz[...] = pid, xyz
                pid, xyz = ppid, pxyz
            o = dict(kf_keep=np.zeros(max(1, len(kf_ids)), np.int32), n_out=np.zeros(1, np.int32), ringkey=np.zeros(num_r, np.float32),
                     sig_idx=np.zeros(num_s * num_r, np.int32), sig_val=np.zeros(num_s * num_r), n_sig=np.zeros(1, np.int32), tfm=np.zeros(16))
            if selected_points:
                o.update(sel_idx=np.zeros(max(1, len(pid)), np.int32), pts_spherical=np.zeros((max(1, len(pid)), 3)))
            self.keepalive.append((kf_ids, poses, cw, pid, xyz))
            self.outs.append((o, len(kf_ids)))
            J = self.arr[j]
            J.n_kf, J.kf_ids, J.kf_pose_wc, J.cur_cw = len(kf_ids), kf_ids.ctypes.data_as(ip), poses.ctypes.data_as(dp), cw.ctypes.data_as(dp)
            J.n_pts, J.pt_kf_id, J.pt_xyz = len(pid), pid.ctypes.data_as(ip), xyz.ctypes.data_as(dp)
            J.kf_keep, J.n_out = o["kf_keep"].ctypes.data_as(ip), o["n_out"].ctypes.data_as(ip)
            if selected_points:
                J.sel_idx, J.pts_spherical = o["sel_idx"].ctypes.data_as(ip), o["pts_spherical"].ctypes.data_as(dp)
            if scancontext:
                J.ringkey, J.sig_idx, J.sig_val = _fp(o["ringkey"]), o["sig_idx"].ctypes.data_as(ip), o["sig_val"].ctypes.data_as(dp)
                J.n_sig, J.tfm_pca_rig = o["n_sig"].ctypes.data_as(ip), o["tfm"].ctypes.data_as(dp)
        self.cand = self.ncand = None
        if db is not None:
            self.cand, self.ncand = np.full((len(jobs), db.k), -1, np.int32), np.zeros(len(jobs), np.int32)

    def run(self):
        ip = c_int_p
        if self.db is None:
            check(self.L.dsm_loop_descriptors_batch(self.ctx.h, len(self.arr), self.arr, self.lidar_range, self.num_s, self.num_r))
        else:
            check(self.L.dsm_loop_detect_batch(self.ctx.h, self.db.h, len(self.arr), self.arr, self.lidar_range, self.num_s, self.num_r,
                                               self.cand.ctypes.data_as(ip), self.ncand.ctypes.data_as(ip)))

    def results(self):
        res = []
        for j, (o, n_kf) in enumerate(self.outs):
            n, ns = int(o["n_out"][0]), int(o["n_sig"][0])
            r = dict(kf_keep=o["kf_keep"][:n_kf].astype(bool), n_out=n)
            if self.selected_points:
                r.update(sel_idx=o["sel_idx"][:n].copy(), pts_spherical=o["pts_spherical"][:n].copy())
            if self.cand is not None:
                r["candidates"] = [int(c) for c in self.cand[j, : self.ncand[j]]]
            if self.scancontext:
                r.update(ringkey=o["ringkey"].copy(), sig_idx=o["sig_idx"][:ns].copy(), sig_val=o["sig_val"][:ns].copy(), tfm_pca_rig=o["tfm"].reshape(4, 4).copy())
            res.append(r)
        return res


def loop_descriptors_batch(ctx, jobs, lidar_range, num_s=60, num_r=20, scancontext=True, db=None, selected_points=True, pinned_clouds=False):
    """DEVICE form of generate_spherical_points + ScanContext::generate for a batch of keyframes (dsm_loop_descriptors_batch).
    jobs: list of (kf_ids, kf_pose_wc, cur_cw, pt_kf_id, pt_xyz).  Returns, per job, a dict with kf_keep, sel_idx,
    pts_spherical and -- with scancontext -- ringkey, sig_idx, sig_val, tfm_pca_rig.
    db (a RingKeyDB): dsm_loop_detect_batch instead -- the jobs' ring keys are searched in (and enqueued into) the index on the device,
    one enqueue and one read-back for the whole chain; every result also carries `candidates` (search_ringkey's list).
    selected_points = False: sel_idx / pts_spherical stay on the device (NULL outputs)."""
    b = LoopBatch(ctx, jobs, lidar_range, num_s, num_r, scancontext, db, selected_points, pinned_clouds)
    b.run()
    return b.results()


def write_trajectory(path, incoming_ids, t_wc):
    """dslam.txt / sodso.txt writer of LoopHandler::savePose (LoopHandler.cpp:59-80) through the C ABI"""
    ids = np.ascontiguousarray(incoming_ids, np.int32)
    t = np.ascontiguousarray(t_wc, np.float64).reshape(-1, 3)
    assert len(ids) == len(t)
    check(_lib.load().dsm_write_trajectory(str(path).encode(), len(ids), ids.ctypes.data_as(c_int_p), t.ctypes.data_as(_lib.c_double_p)))
