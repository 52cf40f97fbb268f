"""Cross-shard merge of the sharded ring-key DB behind the C ABI (csrc/comm_capi.hip; SURVEY.md section 8e;
the k-NN of search_place.h:29-33 over an index split `ordinal mod G`).

One GPU is enough: G "ranks" run as threads, each with its own context and shard on the same device, and the collective of
dsm_ringdb_merge_topk_with is a barrier-synchronised exchange through host memory -- the merge KERNELS and their round
logic are exactly what dsm_ringdb_merge_topk runs over RCCL.  The RCCL binding itself is exercised with a communicator of
one rank (RCCL refuses two ranks on one device); N > 1 ranks over xGMI run in bench.py --gpus N, which checks the merged
result against the unsharded answer on every rank."""
import threading

import numpy as np
import pytest
import torch

from direct_stereo_slam_amd.ringdb import Comm, RingKeyDB
from direct_stereo_slam_amd.tracker import Context
from test_oracle_ringkey import ring_keys

pytestmark = pytest.mark.gpu


class HostExchange:
    """all-reduce(min) / all-gather among G threads through host memory"""

    def __init__(self, G):
        self.G, self.barrier, self.slots = G, threading.Barrier(G), [None] * G

    def allreduce_min(self, rank):
        def fn(buf, count, stream):
            torch.cuda.synchronize()
            h = torch.empty(count, dtype=torch.int64)
            _memcpy(h.data_ptr(), buf, 8 * count, "d2h")
            self.slots[rank] = h
            self.barrier.wait()
            m = torch.stack(self.slots).min(0).values.contiguous()  # packed candidates are non-negative: int64 order = uint64 order
            self.barrier.wait()
            _memcpy(buf, m.data_ptr(), 8 * count, "h2d")
        return fn

    def allgather(self, rank):
        def fn(send, recv, count, stream):
            torch.cuda.synchronize()
            h = torch.empty(count, dtype=torch.int64)
            _memcpy(h.data_ptr(), send, 8 * count, "d2h")
            self.slots[rank] = h
            self.barrier.wait()
            allv = torch.cat(self.slots).contiguous()
            self.barrier.wait()
            _memcpy(recv, allv.data_ptr(), 8 * count * self.G, "h2d")
        return fn


_HIP = None


def _hip():
    """the HIP runtime this process already uses (the copy torch / the product library mapped), not a second one"""
    global _HIP
    if _HIP is None:
        import ctypes as C

        paths = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
        _HIP = C.CDLL(paths[0] if paths else "libamdhip64.so")
    return _HIP


def _memcpy(dst, src, nbytes, kind):
    import ctypes as C

    hip = _hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rc = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), nbytes, 2 if kind == "d2h" else 1)
    assert rc == 0, rc


def _data(n=5000, nq=40, seed=7):
    keys = ring_keys(n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    q = (keys[rng.integers(n, size=nq)] + rng.normal(0, 0.02, (nq, 20))).astype(np.float32)
    q[-1] = 5.0  # a query with no candidate under the threshold: NO_CANDIDATE rows must survive the merge
    return keys, q


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("algo", ["allreduce_min", "allgather"])
def test_merge_kernels_over_G_ranks_equal_the_unsharded_scan(built, G, algo):
    keys, q = _data()
    ctx0 = Context(0)
    full = RingKeyDB(ctx0, capacity=len(keys) + 16)
    full.add_points(keys)
    want = full.knn_packed_host(q)
    ex = HostExchange(G)
    got, errs = [None] * G, []

    def rank_main(r):
        try:
            ctx = Context(0)
            db = RingKeyDB(ctx, capacity=len(keys) // G + 16, shard_rank=r, shard_count=G)
            db.add_points(keys)
            local = torch.from_numpy(db.knn_packed_host(q)).cuda()
            db.merge_topk_with(local.data_ptr(), len(q), G, allreduce_min=ex.allreduce_min(r), allgather=ex.allgather(r), algo=algo)
            ctx.sync()
            got[r] = local.cpu().numpy()
            db.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for r in range(G):
        assert np.array_equal(got[r], want), f"rank {r}"
    full.close()
    ctx0.close()


def test_rccl_binding_with_one_rank(built, ctx):
    """librccl is loaded by the library itself; a one-rank communicator runs the k all-reduce rounds / the all-gather"""
    keys, q = _data(n=2000, nq=16)
    comm = Comm(ctx, Comm.unique_id(), 0, 1)
    db = RingKeyDB(ctx, capacity=len(keys) + 16)
    db.add_points(keys)
    want = db.knn_packed_host(q)
    for algo in ("allreduce_min", "allgather"):
        local = torch.from_numpy(want.copy()).cuda()
        db.merge_topk_device(comm, local.data_ptr(), len(q), algo)
        ctx.sync()
        assert np.array_equal(local.cpu().numpy(), want)
    db.close()
    comm.close()


def test_sharded_query_then_enqueue_needs_a_communicator(built, ctx):
    from direct_stereo_slam_amd._lib import DsmError

    db = RingKeyDB(ctx, shard_rank=0, shard_count=2)
    db.add_points(ring_keys(50, seed=1))
    with pytest.raises(DsmError, match="communicator"):
        db.search_ringkey(np.zeros(20, np.float32))
    comm = Comm(ctx, Comm.unique_id(), 0, 1)
    with pytest.raises(DsmError, match="shard"):
        db.attach_comm(comm)  # rank / size of the communicator must equal the shard's
    comm.close()
    db.close()


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_search_ringkey_is_a_drop_in_for_the_unsharded_one(built, G):
    """search_ringkey (search_place.h:25-57) on a sharded index: every rank calls dsm_ringdb_query_then_enqueue with the same
    key -- scan of its shard, cross-shard merge, delay queue -- and gets the candidate list the unsharded index returns,
    query after query while the index grows through the delay queue (margin 5 here) and its dummy slot"""
    keys = ring_keys(160, seed=11)
    keys[40:60] = keys[:20] + np.float32(0.01)  # revisits: candidates under the threshold
    keys[100:130] = keys[30:60]
    ctx0 = Context(0)
    ref_db = RingKeyDB(ctx0, margin=5, capacity=64)
    want = [ref_db.search_ringkey(k) for k in keys]
    assert any(len(w) for w in want)
    ex = HostExchange(G)
    got, errs = [None] * G, []

    def rank_main(r):
        try:
            ctx = Context(0)
            db = RingKeyDB(ctx, margin=5, capacity=64, shard_rank=r, shard_count=G)
            db.attach_transport(G, ex.allreduce_min(r))
            got[r] = [db.search_ringkey(k) for k in keys]
            assert db.size() == ref_db.size()
            db.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for r in range(G):
        assert got[r] == want, f"rank {r}"
    ref_db.close()
    ctx0.close()


def test_collective_query_fails_on_every_rank_when_one_rank_cannot_take_part(built):
    """round 0 of a collective query is an agreement: a rank whose index differs (or whose local scan failed) makes EVERY rank
    return an error -- nobody is left waiting in the merge rounds"""
    from direct_stereo_slam_amd._lib import DsmError

    G = 3
    keys = ring_keys(60, seed=3)
    ex = HostExchange(G)
    outcome, errs = [None] * G, []

    def rank_main(r):
        try:
            ctx = Context(0)
            db = RingKeyDB(ctx, margin=5, capacity=64, shard_rank=r, shard_count=G)
            db.attach_transport(G, ex.allreduce_min(r))
            db.add_points(keys[:40] if r != 1 else keys[:41])  # rank 1 holds one entry more
            try:
                db.search_ringkey(keys[50])
                outcome[r] = "ok"
            except DsmError as e:
                outcome[r] = str(e)
            db.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs and not any(t.is_alive() for t in ts), errs
    assert all(o is not None and "different numbers of entries" in o for o in outcome), outcome


def test_destroying_a_communicator_detaches_it(built, ctx):
    from direct_stereo_slam_amd._lib import DsmError

    db = RingKeyDB(ctx)
    comm = Comm(ctx, Comm.unique_id(), 0, 1)
    db.attach_comm(comm)
    comm.close()  # the database outlives it: detached, not dangling
    db.add_points(ring_keys(20, seed=2))
    assert db.search_ringkey(np.zeros(20, np.float32)) == []  # one shard: no collective needed
    db2 = RingKeyDB(ctx, shard_rank=0, shard_count=2)
    with pytest.raises(DsmError, match="communicator"):
        db2.search_ringkey(np.zeros(20, np.float32))
    db.close()
    db2.close()


_TWO_RANKS = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from direct_stereo_slam_amd.ringdb import Comm, RingKeyDB
from direct_stereo_slam_amd.tracker import Context
from test_oracle_ringkey import ring_keys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo")  # plumbing only (the unique id); the merge itself runs on RCCL through the C ABI
ctx = Context(rank)
uid = [Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, 0)
comm = Comm(ctx, uid[0], rank, world)
keys = ring_keys(5000, seed=7)
rng = np.random.default_rng(8)
q = (keys[rng.integers(len(keys), size=40)] + rng.normal(0, 0.02, (40, 20))).astype(np.float32)
q[-1] = 5.0
full = RingKeyDB(ctx, capacity=len(keys) + 16); full.add_points(keys); want = full.knn_packed_host(q)
db = RingKeyDB(ctx, capacity=len(keys) // world + 16, shard_rank=rank, shard_count=world); db.add_points(keys)
for algo in ("allreduce_min", "allgather"):
    local = torch.from_numpy(db.knn_packed_host(q)).cuda()
    db.merge_topk_device(comm, local.data_ptr(), len(q), algo); ctx.sync()
    assert np.array_equal(local.cpu().numpy(), want), (rank, algo)
# search_ringkey as a collective over xGMI: the candidate lists of the unsharded index, query after query
ref = RingKeyDB(ctx, margin=5, capacity=64); sh = RingKeyDB(ctx, margin=5, capacity=64, shard_rank=rank, shard_count=world)
sh.attach_comm(comm)
ks = ring_keys(120, seed=11); ks[40:60] = ks[:20] + np.float32(0.01)
assert [sh.search_ringkey(k) for k in ks] == [ref.search_ringkey(k) for k in ks]
dist.barrier()
if rank == 0: print("RCCL-2-RANKS-OK")
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_rccl_merge_with_two_ranks_over_xgmi(built, tmp_path):
    """the first multi-GPU box runs dsm_ringdb_merge_topk and the collective search_ringkey over real RCCL ranks"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "two_ranks.py")
    open(script, "w").write(_TWO_RANKS)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29531", script, root], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0 and "RCCL-2-RANKS-OK" in res.stdout, res.stderr[-3000:]
