"""CPU: the brute-force ring-key oracle against scipy / numpy, and the delay-queue semantics of
search_ringkey (search_place.h:25-57)."""
import numpy as np
from scipy.spatial import cKDTree

from oracle import oracle as O


def ring_keys(n, seed=1234):
    """entry = Binomial(60, p_ring)/60 with p_ring ~ U(0.1,0.9) per ring"""
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.1, 0.9, 20)
    return (rng.binomial(60, p, size=(n, 20)) / 60.0).astype(np.float32)


def test_l2_matches_numpy():
    rng = np.random.default_rng(0)
    a, b = rng.uniform(size=20).astype(np.float32), rng.uniform(size=20).astype(np.float32)
    d = O.lib().orc_l2_sq(a.ctypes.data_as(O.c_float_p), b.ctypes.data_as(O.c_float_p), 20)
    assert abs(d - ((a.astype(np.float64) - b) ** 2).sum()) < 1e-6


def test_knn_matches_kdtree():
    keys = ring_keys(3000)
    db = O.OracleRingDB(thres=np.inf, dummy=np.full(20, 5.0, np.float32))  # dummy far away
    db.add_points(keys)
    tree = cKDTree(np.vstack([np.full((1, 20), 5.0), keys.astype(np.float64)]))
    rng = np.random.default_rng(3)
    for _ in range(50):
        q = (keys[rng.integers(len(keys))] + rng.normal(0, 0.02, 20)).astype(np.float32)
        idx, dist = db.knn(q)
        dd, ii = tree.query(q.astype(np.float64), k=3)
        np.testing.assert_allclose(np.sqrt(dist), dd, rtol=1e-5)
        if len(set(np.round(dd, 9))) == 3:  # no exact ties
            assert idx == list(ii)


def test_tie_break_smaller_index_first():
    db = O.OracleRingDB(thres=np.inf, dummy=np.ones(20, np.float32))
    k = np.zeros((4, 20), np.float32)
    k[:, 0] = [0.5, 0.25, 0.25, 0.25]
    db.add_points(k)
    idx, dist = db.knn(np.zeros(20, np.float32))
    assert idx == [2, 3, 4] and dist[0] == dist[1] == dist[2]


def test_delay_queue_and_dummy_semantics():
    keys = ring_keys(400, seed=7)
    db = O.OracleRingDB(dim=20, margin=100, k=3, thres=0.1)
    out = []
    for i, k in enumerate(keys):
        assert db.size() == 1 + max(0, i - 100)  # key i enters the index only LOOP_MARGIN calls later (:46-52)
        out.append(db.query_then_enqueue(k))
    # the first 103 queries cannot return anything: index size must exceed FLANN_NN (:29)
    assert all(len(c) == 0 for c in out[:103])
    # candidate ordinals refer to searched frames at least LOOP_MARGIN calls ago
    for i, c in enumerate(out):
        for ordinal in c:
            assert 0 <= ordinal < i - 99
    assert any(len(c) > 0 for c in out)
    # candidates are in ascending distance order and below the threshold
    i = max(j for j, c in enumerate(out) if len(c) >= 2)
    d = [((keys[i] - keys[o]) ** 2).sum() for o in out[i]]
    assert d == sorted(d) and d[-1] < 0.1


def test_dummy_can_occupy_a_slot():
    # quirk Q8: index slot 0 is a dummy that can take one of the k result slots, then is dropped
    dummy = np.full(20, 0.5, np.float32)
    db = O.OracleRingDB(thres=0.1, dummy=dummy)
    near = np.tile(dummy, (5, 1)) + np.linspace(0.01, 0.05, 5, dtype=np.float32)[:, None] * np.eye(20, dtype=np.float32)[0]
    db.add_points(near)
    idx, _ = db.knn(dummy)
    assert idx[0] == 0  # dummy itself is the nearest
    # emulate the caller-visible filter
    db2 = O.OracleRingDB(thres=0.1, dummy=dummy, margin=1)
    for k in near:
        db2.query_then_enqueue(k)
    cands = db2.query_then_enqueue(dummy)
    assert len(cands) == 2  # 3 nearest = dummy + 2 keys; dummy dropped, not replaced
