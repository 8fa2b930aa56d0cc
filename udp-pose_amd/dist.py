"""One-process-per-GPU helpers over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

Inference shards naturally: every person crop is independent, so ranks take contiguous slices of
the box list (the reference's multi-GPU test does the same, RSN/lib/utils/dataloader.py:87-92),
run the whole hot path on their slice with NO data-path collective, and exchange only the final
fixed-shape ``[n_r, J, 3]`` keypoint records (the reference all_gathers pickled lists instead,
RSN/lib/utils/comm.py:47-87).

Training has one real exchange step per iteration: the gradient all-reduce that nn.DataParallel
performs implicitly in the reference (deep_hrnet/tools/train.py:116 with lib/core/function.py:66-77):
sum over ranks, divide by world size == loss averaged over the global batch for equal shards.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world_size):
    """Contiguous [lo, hi) slice of n items for `rank` (first n % world_size ranks get one more)."""
    if world_size < 1 or not 0 <= rank < world_size or n < 0:
        raise ValueError("bad shard request n=%d rank=%d world=%d" % (n, rank, world_size))
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_keypoints(local, n_total, group=None):
    """All ranks end up with the [n_total, J, K] tensor made of every rank's contiguous shard.

    `local` is this rank's [n_r, J, K] result (any device the backend supports).  Shards are padded
    to the largest shard so a single fixed-shape all_gather suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d rows, shard is [%d,%d)" % (rank, local.shape[0], lo, hi))
    cap = -(-n_total // world)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        a, b = shard_bounds(n_total, r, world)
        out.append(parts[r][: b - a])
    return torch.cat(out, dim=0)


def allreduce_sum_(flat_grads, bucket_elems=6 * 1024 * 1024, group=None):
    """In-place SUM over ranks of a flat gradient buffer, in ~25 MB buckets issued back to back (xGMI is
    point-to-point: several buckets in flight keep every link busy).  The 1/world_size of the mean is
    applied by the consumer (udp_adam_step's grad_scale): no arithmetic outside RCCL and our kernels."""
    works = [dist.all_reduce(flat_grads[lo:lo + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True)
             for lo in range(0, flat_grads.numel(), bucket_elems)]
    for w in works:
        w.wait()
    return works


def allreduce_sum_async(bucket, group=None):
    """SUM all-reduce of one gradient bucket, not waited for: the returned work handle's ``wait()`` orders the
    caller's stream after it.  torch.distributed enqueues the collective on its own stream behind everything
    already queued on the current stream, so it may be issued right after the kernels that wrote the bucket and
    runs concurrently with the kernels issued afterwards (the rest of the backward)."""
    return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_mean_(flat_grads, bucket_elems=6 * 1024 * 1024, group=None):
    """In-place mean over ranks of a flat gradient buffer, in buckets (xGMI is point-to-point:
    ~25 MB fp32 buckets keep every link busy and let later buckets overlap remaining backward work).
    Returns the list of async work handles already waited on."""
    world = dist.get_world_size(group)
    works = []
    for lo in range(0, flat_grads.numel(), bucket_elems):
        works.append(dist.all_reduce(flat_grads[lo:lo + bucket_elems], op=dist.ReduceOp.SUM, group=group,
                                     async_op=True))
    for w in works:
        w.wait()
    flat_grads.div_(world)
    return works
