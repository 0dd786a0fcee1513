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


def allreduce_flat_(gflat: torch.Tensor, group=None, bucket_elems: int = 0, single_rank: bool = False):
    """In-place SUM all-reduce of the flat gradient buffer. bucket_elems > 0 splits it into contiguous buckets
    (issued back to back on the collective stream; one bucket = one collective). Returns the world size, i.e. the
    factor the caller divides by (folded into the optimiser kernel as grad_scale = 1 / world)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1 and not single_rank:
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


def broadcast_flat_(pflat: torch.Tensor, src: int = 0, group=None, single_rank: bool = False):
    """make every rank start from rank `src`'s weights (one collective over the flat parameter buffer)"""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank):
        dist.broadcast(pflat, src=src, group=group)


class GradExchange:
    """The per-step gradient exchange of the data-parallel train step, device-agnostic (RCCL on the GPU, gloo in the CPU tests).

    `ranges`: ordered dict tag -> (lo, hi) element range of the flat f32 gradient buffer, in the order the backward pass finishes
    them.  overlap=True: `bucket_ready(tag)` launches that bucket's SUM all-reduce asynchronously (on the collective's own stream)
    and `finish()` only waits; overlap=False: `finish()` runs ONE all-reduce over the whole buffer.
    comm_dtype=torch.bfloat16 halves the bytes on the wire: a bucket is cast into a bf16 staging buffer, reduced there, and
    written back to the f32 buffer (each rank's contribution is rounded to 8 significant bits and the sum is formed in bf16:
    relative error of the reduced gradient ~2^-8 * sqrt(world); the optimiser state and the weights stay f32).
    The caller folds 1/world into its update."""

    def __init__(self, ranges, group=None, overlap=True, comm_dtype=torch.float32, single_rank_collectives=False):
        """single_rank_collectives: issue the collectives on a 1-rank group too (they are skipped otherwise) -- the whole RCCL
        code path, its streams and staging buffers included, then runs on one GPU (tests/test_gpu_ddp.py)"""
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('comm_dtype must be torch.float32 or torch.bfloat16')
        self.ranges, self.group, self.overlap, self.comm_dtype = dict(ranges), group, overlap, comm_dtype
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.active = self.world > 1 or (single_rank_collectives and dist.is_available() and dist.is_initialized())
        self._buf = None
        self._works, self._launched, self._g = [], set(), None

    def begin(self, gflat):
        """arm for one backward pass over `gflat`"""
        self._g, self._works, self._launched = gflat, [], set()

    def _stage(self, gflat):
        if self._buf is None or self._buf.numel() != gflat.numel() or self._buf.device != gflat.device:
            self._buf = torch.empty(gflat.numel(), dtype=self.comm_dtype, device=gflat.device)
        return self._buf

    def bucket_ready(self, tag):
        if not self.active or not self.overlap or tag not in self.ranges or tag in self._launched:
            return
        lo, hi = self.ranges[tag]
        self._launched.add(tag)
        if self.comm_dtype == torch.float32:
            src = self._g[lo:hi]
        else:   # stream-ordered after the kernels that produced the bucket
            src = self._stage(self._g)[lo:hi]
            src.copy_(self._g[lo:hi])
        self._works.append((dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi))

    def finish(self):
        """after the backward pass: every element of the gradient buffer holds the SUM over ranks when this returns"""
        if not self.active:
            return
        g = self._g
        if self.overlap:
            missing = set(self.ranges) - self._launched
            if missing:
                raise RuntimeError(f'gradient buckets never reported ready: {sorted(missing)}')
            for w, lo, hi in self._works:
                w.wait()
                if self.comm_dtype != torch.float32:
                    g[lo:hi].copy_(self._buf[lo:hi])
            self._works = []
        elif self.comm_dtype != torch.float32:
            buf = self._stage(g)
            buf.copy_(g)
            allreduce_flat_(buf, group=self.group, single_rank=True)
            g.copy_(buf)
        else:
            allreduce_flat_(g, group=self.group, single_rank=True)
