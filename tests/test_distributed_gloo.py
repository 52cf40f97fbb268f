"""CPU, world_size 2, gloo: the cross-shard merge of the ring-key DB (the only collective on the
path).  The per-shard k-NN is produced here by a numpy brute force with the oracle's distance
function, so that the test needs no GPU; on the GPU box the same merge runs over RCCL with the
kernel's output (bench.py --gpus N)."""
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
    from direct_stereo_slam_amd.ringdb import merge_topk_allreduce_min
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


def test_allreduce_min_merge_world2(built):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
