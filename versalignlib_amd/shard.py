"""Pair-range sharding across ranks (one process per GPU) and the score all-gather.

Pairs are independent -- the reference's only parallelism is a loop over them
(src/Kernels/default/DefaultKernel.cpp:45-48, 76-79) -- so a batch shards into contiguous
ranges with no exchange during the DP.  The one collective is the all-gather of the
per-shard int16 scores (RCCL over xGMI on GPUs, gloo in the CPU tests).  NCCL/RCCL has no
int16 datatype; the gather moves bits only, so scores travel as bytes.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous range [begin, end) of rank: ceil(n/world) pairs each, last ones short."""
    per = -(-int(n) // int(world)) if world > 0 else int(n)
    begin = min(int(n), rank * per)
    return begin, min(int(n), begin + per)


def shard_sizes(n, world):
    return [e - b for b, e in (shard_range(n, r, world) for r in range(world))]


def all_gather_scores(local_scores, n_total=None, group=None):
    """All ranks contribute an int16 tensor [n_local]; every rank gets the concatenation.

    Shards may differ in length (tail shard): they are padded to the longest for the
    collective and trimmed afterwards.  With n_total given, the lengths are the
    shard_sizes(n_total, world) and no size exchange is needed.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_scores
    assert local_scores.dtype == torch.int16 and local_scores.dim() == 1
    dev = local_scores.device
    if n_total is not None:
        sizes = shard_sizes(n_total, world)
    else:
        mine = torch.tensor([local_scores.numel()], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        sizes = [int(t.item()) for t in every]
    longest = max(sizes) if sizes else 0
    send = local_scores
    if local_scores.numel() != longest:
        send = torch.zeros(longest, dtype=torch.int16, device=dev)
        send[:local_scores.numel()] = local_scores
    send_bytes = send.contiguous().view(torch.uint8)
    recv = torch.empty(world * longest * 2, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send_bytes, group=group)
    full = recv.view(torch.int16).view(world, longest)
    if all(s == longest for s in sizes):
        return full.reshape(-1)
    return torch.cat([full[r, :sizes[r]] for r in range(world)])


def score_sharded(score_fn, reads, refs, group=None):
    """Score the global batch (every rank holds it, or at least its own range) by shards.

    score_fn(reads_shard, refs_shard) -> int16 tensor [n_shard] on the shard's device.
    Returns the full score vector on every rank.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = reads.shape[0]
    b, e = shard_range(n, rank, world)
    local = score_fn(reads[b:e], refs[b:e])
    return all_gather_scores(local, n_total=n, group=group)
