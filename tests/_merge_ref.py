"""Torch restatement of the sharded ring-key merge (k rounds of all-reduce(min) over the shards' heads, the
owner of a round's winner pops it) -- TEST INFRASTRUCTURE: the product's merge is dsm_ringdb_merge_topk (csrc/comm_capi.hip);
this is what tests/test_parity_ringkey.py and tests/test_distributed_gloo.py hold it against."""
from direct_stereo_slam_amd.ringdb import NO_CANDIDATE


def merge_topk_allreduce_min(local_sorted, k, all_reduce_min):
    """local_sorted: (nq, k) int64 torch tensor, ascending per row, NO_CANDIDATE padded.
    all_reduce_min(tensor) performs the in-place element-wise MIN all-reduce over the shards.
    Returns the global (nq, k) top-k, identical on every rank."""
    import torch

    nq = local_sorted.shape[0]
    ptr = torch.zeros(nq, dtype=torch.int64, device=local_sorted.device)
    padded = torch.cat([local_sorted, torch.full((nq, 1), NO_CANDIDATE, dtype=torch.int64, device=local_sorted.device)], 1)
    out = torch.empty((nq, k), dtype=torch.int64, device=local_sorted.device)
    for r in range(k):
        head = padded.gather(1, ptr[:, None])[:, 0].contiguous()
        gmin = head.clone()
        all_reduce_min(gmin)
        out[:, r] = gmin
        ptr = ptr + ((head == gmin) & (gmin != NO_CANDIDATE)).to(torch.int64)  # indices are unique: one owner pops
    return out
