"""world_size 2, gloo: the cross-shard merge of the ring-key DB (the only collective on the path).

Without a GPU (`-m "not gpu"`): the per-shard k-NN is a numpy brute force with the oracle's distance function and the merge
is the torch restatement of the round algorithm (tests/_merge_ref.py: merge_topk_allreduce_min).
With a GPU (`-m gpu`): two gloo processes share the device; each scans ITS shard with the HIP kernel
(dsm_ringdb_knn_packed_host) and the merge is the C ABI's own (dsm_ringdb_merge_topk_with: the kernels RCCL drives in
dsm_ringdb_merge_topk) with gloo as the transport.  N ranks over RCCL / xGMI: bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO = 0x7FFFFFFFFFFFFFFF


def pack_topk(keys_global_idx, keys, q, k, thres):
    from oracle import oracle as O

    L = O.lib()
    out = np.full((len(q), k), NO, np.int64)
    for i, qq in enumerate(q):
        cands = []
        for g, key in zip(keys_global_idx, keys):
            d = np.float32(L.orc_l2_sq(qq.ctypes.data_as(O.c_float_p), key.ctypes.data_as(O.c_float_p), 20))
            if d < thres:
                cands.append((int(d.view(np.uint32)) << 32) | int(g))
        cands.sort()
        out[i, : min(k, len(cands))] = cands[:k]
    return out


def worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _merge_ref import merge_topk_allreduce_min
    from test_oracle_ringkey import ring_keys

    keys = ring_keys(600, seed=5)
    rng = np.random.default_rng(2)
    q = (keys[rng.integers(600, size=24)] + rng.normal(0, 0.02, (24, 20))).astype(np.float32)
    glob = np.arange(len(keys))
    mine = glob % world == rank  # ordinal mod G sharding
    local = torch.from_numpy(pack_topk(glob[mine], np.ascontiguousarray(keys[mine]), q, 3, 0.1))
    merged = merge_topk_allreduce_min(local, 3, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MIN))
    full = pack_topk(glob, keys, q, 3, 0.1)
    ok = bool((merged.numpy() == full).all())
    gathered = [torch.zeros_like(merged) for _ in range(world)]
    dist.all_gather(gathered, merged)
    same = all(bool((g == merged).all()) for g in gathered)
    ret[rank] = ok and same
    dist.barrier()
    dist.destroy_process_group()


def _run(fn):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_allreduce_min_merge_world2(built):
    _run(worker)


def gpu_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from direct_stereo_slam_amd.ringdb import RingKeyDB
    from direct_stereo_slam_amd.tracker import Context
    from test_comm_merge import _memcpy
    from test_oracle_ringkey import ring_keys

    keys = ring_keys(3000, seed=5)
    rng = np.random.default_rng(2)
    q = (keys[rng.integers(3000, size=24)] + rng.normal(0, 0.02, (24, 20))).astype(np.float32)
    ctx = Context(0)
    db = RingKeyDB(ctx, capacity=len(keys) // world + 16, shard_rank=rank, shard_count=world)  # ordinal mod G shard
    db.add_points(keys)

    def allreduce_min(buf, count, stream):
        torch.cuda.synchronize()
        h = torch.empty(count, dtype=torch.int64)
        _memcpy(h.data_ptr(), buf, 8 * count, "d2h")
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        _memcpy(buf, h.data_ptr(), 8 * count, "h2d")

    def allgather(send, recv, count, stream):
        torch.cuda.synchronize()
        h = torch.empty(count, dtype=torch.int64)
        _memcpy(h.data_ptr(), send, 8 * count, "d2h")
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        allv = torch.cat(parts).contiguous()
        _memcpy(recv, allv.data_ptr(), 8 * count * world, "h2d")

    ok = True
    full = RingKeyDB(ctx, capacity=len(keys) + 16)
    full.add_points(keys)
    want = full.knn_packed_host(q)
    for algo in ("allreduce_min", "allgather"):
        local = torch.from_numpy(db.knn_packed_host(q)).cuda()  # the HIP kernel's scan of this shard
        db.merge_topk_with(local.data_ptr(), len(q), world, allreduce_min=allreduce_min, allgather=allgather, algo=algo)
        ctx.sync()
        ok = ok and bool((local.cpu().numpy() == want).all())
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_abi_merge_of_kernel_shards_over_gloo_world2(built):
    _run(gpu_worker)
