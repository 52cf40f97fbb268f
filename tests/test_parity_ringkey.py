"""GPU parity: ring-key search through the C ABI against the brute-force oracle.  Candidate
indices must be bit-exact, distances too (same float operation order)."""
import numpy as np
import pytest

from direct_stereo_slam_amd.ringdb import RingKeyDB, candidates_from_packed, unpack
from oracle import oracle as O

from test_oracle_ringkey import ring_keys

pytestmark = pytest.mark.gpu


def test_search_ringkey_sequence_bit_exact(ctx):
    """500-key replay with the LOOP_MARGIN delay queue (SURVEY.md section 8c fixture shape)"""
    keys = ring_keys(500, seed=99)
    rng = np.random.default_rng(1)
    # revisit earlier places so that candidates exist
    for i in range(150, 500, 7):
        keys[i] = keys[i - 130] + (rng.integers(-1, 2, 20) / 60.0).astype(np.float32) * (rng.uniform(size=20) < 0.2)
    dummy = np.full(20, 0.5, np.float32)
    orc = O.OracleRingDB(dummy=dummy)
    db = RingKeyDB(ctx, dummy=dummy, capacity=64)  # small capacity: exercises the growth path
    n_with = 0
    for k in keys:
        co, cg = orc.query_then_enqueue(k), db.search_ringkey(k)
        assert cg == co
        n_with += bool(co)
        assert db.size() == orc.size()
    assert n_with > 20


@pytest.mark.parametrize("n,nq", [(5, 3), (129, 17), (5000, 300), (200000, 64)])
def test_knn_packed_matches_oracle(ctx, n, nq):
    keys = ring_keys(n, seed=n)
    rng = np.random.default_rng(n)
    q = (keys[rng.integers(n, size=nq)] + rng.normal(0, 0.02, (nq, 20))).astype(np.float32)
    q[0] = keys[min(3, n - 1)]  # exact hit: distance 0
    for thres in (0.1, np.inf):
        orc = O.OracleRingDB(thres=thres)
        orc.add_points(keys)
        db = RingKeyDB(ctx, thres=thres)
        db.add_points(keys)
        packed = db.knn_packed_host(q)
        dist, idx = unpack(packed)
        for i in range(min(nq, 40)):
            io, do = orc.knn(q[i])
            exp = [(d, j) for d, j in zip(do, io) if j >= 0 and d < thres]
            got = [(d, j) for d, j in zip(dist[i], idx[i]) if j >= 0]
            assert [j for _, j in got] == [j for _, j in exp]
            assert [np.float32(d) for d, _ in got] == [np.float32(d) for d, _ in exp]  # bit exact distances


def test_ties_and_duplicates(ctx):
    keys = np.tile(ring_keys(1, seed=5), (700, 1))  # 700 identical keys: pure tie-break
    db = RingKeyDB(ctx, thres=np.inf, dummy=np.full(20, 9.0, np.float32))
    db.add_points(keys)
    dist, idx = unpack(db.knn_packed_host(keys[:1]))
    assert list(idx[0]) == [1, 2, 3] and np.all(dist[0] == 0)


def test_sharded_db_merges_to_unsharded_result(ctx):
    """two shards on one GPU, merged on the host with the same k-round min-with-pop protocol the
    RCCL path uses (tests/_merge_ref.py: merge_topk_allreduce_min)"""
    import torch

    from _merge_ref import merge_topk_allreduce_min

    keys = ring_keys(3001, seed=77)
    rng = np.random.default_rng(0)
    q = (keys[rng.integers(3001, size=50)] + rng.normal(0, 0.01, (50, 20))).astype(np.float32)
    full = RingKeyDB(ctx)
    full.add_points(keys)
    ref = full.knn_packed_host(q)
    for G in (2, 3, 8):
        shards = [RingKeyDB(ctx, shard_rank=r, shard_count=G) for r in range(G)]
        for s in shards:
            s.add_points(keys)  # every rank sees every key and keeps its own ordinals
            assert s.size() == full.size()
        locs = [torch.from_numpy(s.knn_packed_host(q)) for s in shards]

        # emulate all-reduce(min) over the G ranks
        class Bus:
            def __init__(self):
                self.bufs = []

        merged = []
        # run the protocol for all ranks in lock step
        ptrs = [torch.zeros(50, dtype=torch.int64) for _ in range(G)]
        pad = [torch.cat([l, torch.full((50, 1), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64)], 1) for l in locs]
        out = torch.empty((50, 3), dtype=torch.int64)
        for r in range(3):
            heads = [pad[g].gather(1, ptrs[g][:, None])[:, 0] for g in range(G)]
            gmin = torch.stack(heads).min(0).values
            out[:, r] = gmin
            for g in range(G):
                ptrs[g] += ((heads[g] == gmin) & (gmin != 0x7FFFFFFFFFFFFFFF)).long()
        np.testing.assert_array_equal(out.numpy(), ref)
        # and the library helper gives the same with a 1-rank "all-reduce" on the unsharded result
        single = merge_topk_allreduce_min(torch.from_numpy(ref), 3, lambda t: t)
        np.testing.assert_array_equal(single.numpy(), ref)
        assert [candidates_from_packed(r) for r in out.numpy()] == [candidates_from_packed(r) for r in ref]
