"""
Data-parallel plumbing for the train step (the reference has no distributed code at all).

One process per GPU, launched by `torch.distributed.run`; records are independent in forward/backward (LayerNorm is
per token, BCE 'mean' over equal shards = mean of shard means), so the ONLY exchange per step is the gradient
all-reduce.  All parameters' gradients live in one flat f32 buffer (engine.ParamLayout), so the collective is a
single RCCL all-reduce(sum) over that buffer; the 1/world scaling is folded into the clip+AdamW kernel
(`grad_scale`), and the global-norm clip is computed on the REDUCED gradient, as `clip_grad_norm_` after DDP would.

Device-agnostic on purpose: the world_size-2 `gloo` tests exercise exactly these functions on CPU tensors.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int):
    """contiguous, equal shards (global batch 4096 -> 8 x 512); the path requires divisibility so 'mean' stays exact"""
    if global_batch % world != 0:
        raise ValueError(f'global batch {global_batch} is not divisible by world size {world}')
    per = global_batch // world
    return rank * per, (rank + 1) * per


def rank_seed(base_seed: int, rank: int) -> int:
    """synthetic-input seed per rank (77 + rank)"""
    return base_seed + rank


def allreduce_flat_(gflat: torch.Tensor, group=None, bucket_elems: int = 0):
    """In-place SUM all-reduce of the flat gradient buffer. bucket_elems > 0 splits it into contiguous buckets
    (issued back to back on the collective stream; one bucket = one collective). Returns the world size, i.e. the
    factor the caller divides by (folded into the optimiser kernel as grad_scale = 1 / world)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1:
        return 1
    if bucket_elems <= 0 or bucket_elems >= gflat.numel():
        dist.all_reduce(gflat, op=dist.ReduceOp.SUM, group=group)
    else:
        works = []
        for s in range(0, gflat.numel(), bucket_elems):
            works.append(dist.all_reduce(gflat[s:s + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
    return world


def broadcast_flat_(pflat: torch.Tensor, src: int = 0, group=None):
    """make every rank start from rank `src`'s weights (one collective over the flat parameter buffer)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(pflat, src=src, group=group)
