"""
Train-step contract of the reference trainer, on the HIP engine.

Mirrors `ecg_transformer/models/train.py`:
  get_train_args (:407-436)  -- same defaults, same `steps_per_epoch = ceil(n_train // bsz)` floor-first quirk (:433)
  the optimiser / scheduler wiring of MyTrainer.train (:241-252)
  the step body (:268-283): zero_grad -> forward -> backward -> clip_grad_norm_(1.0, error_if_nonfinite) -> AdamW -> sched

`HipTrainStep` runs that body fused on flat HBM buffers: one sum-of-squares pass over the flat gradient, one
clip+AdamW pass that also refreshes the bf16 weight shadow, no per-parameter launches.  `error_if_nonfinite=True` is
honoured in the SAME step, as the reference does (train.py:281): the kernel skips the whole update on a non-finite norm and
the step reads the flag back before it returns (`sync_nonfinite=False` defers that read by one step to keep the host running
ahead of the device).  Data parallel (reference has none): one process per GPU; the replicas are made identical
once (broadcast of rank 0's flat parameter buffer), every rank draws its own dropout masks (rank folded into the seed), and
gradients are all-reduced over RCCL in per-layer buckets of the flat gradient buffer (28 MB f32 / 14 MB bf16 each for base),
each launched asynchronously the moment the backward pass has finished that layer (head, layers L-1..0, embedding), so the
exchange overlaps the remaining backward.

Logging / TensorBoard / sklearn metrics / datasets of MyTrainer are host-side and out of scope here.
"""
import math
import sys

import torch
import torch.distributed as dist

from .check_args import ca
from . import hip
from . import ddp


def get_train_args(args=None, n_train=None):
    default_args = dict(
        num_train_epoch=3,
        train_batch_size=64,
        eval_batch_size=64,
        do_eval=True,
        optimizer='AdamW',
        learning_rate=3e-4,
        weight_decay=1e-2,
        warmup_ratio=0.05,
        schedule='cosine',
        n_sample=None,
        augment_timeout=False,
        patience=8,
        precision=16 if torch.cuda.is_available() else 'bf16',  # carried, unused by the reference's live trainer too
        log_per_epoch=False,
        log_to_console=True,
        save_every_n_epoch=False,
        save_top_k=-1,
        tqdm=False
    )
    args_ = default_args
    if args is not None:
        args_.update(args)
    args_['steps_per_epoch'] = steps_per_epoch = math.ceil((n_train or int(sys.maxsize)) // args_['train_batch_size'])
    args_['n_step'] = steps_per_epoch * args_['num_train_epoch']
    ca(optimizer=args_['optimizer'], schedule=args_['schedule'])
    return args_


def lr_multiplier(schedule, n_warmup, n_step):
    """HF get_constant_schedule_with_warmup / get_cosine_schedule_with_warmup (num_cycles=0.5) lambdas (train.py:245-252)."""
    ca(schedule=schedule)

    def f(step):
        if step < n_warmup:
            return float(step) / float(max(1, n_warmup))
        if schedule == 'constant':
            return 1.0
        progress = float(step - n_warmup) / float(max(1, n_step - n_warmup))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    return f


class HipTrainStep:
    """
    step(sample_values, labels) == reference train.py:271-283 for one batch, fused.

    args: dict as produced by `get_train_args` (uses optimizer, learning_rate, weight_decay, warmup_ratio, schedule, n_step).
    """

    def __init__(self, model, args=None, max_grad_norm=1.0, sync_nonfinite=True, process_group=None, overlap_allreduce=True,
                 grad_comm_dtype=torch.float32, single_rank_collectives=False):
        """grad_comm_dtype: torch.float32 (exact exchange) or torch.bfloat16 (each bucket is cast to bf16, all-reduced at half the
        bytes over xGMI and added back into the f32 gradient buffer; the optimiser still sees f32).
        single_rank_collectives: run the start broadcast and the gradient exchange on a 1-rank group too (skipped otherwise), so
        that the whole RCCL path -- collective stream, staging buffers, chunked GEMM launches -- can be exercised on one GPU"""
        self.model = model
        self.args = {**get_train_args(), **(args or dict())}
        ca(optimizer=self.args['optimizer'], schedule=self.args['schedule'])
        self.lr0, self.wd = self.args['learning_rate'], self.args['weight_decay']
        n_step = self.args['n_step']
        self.mult = lr_multiplier(self.args['schedule'], round(n_step * self.args['warmup_ratio']), n_step)
        self.decoupled = self.args['optimizer'] == 'AdamW'
        self.max_grad_norm = max_grad_norm
        self.sync_nonfinite = sync_nonfinite
        self.step_count = 0       # optimiser steps taken == scheduler.step() calls
        self.m = self.v = None
        self.norm_out = self.sumsq = self.ws = None
        self.last_loss = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.collectives = self.world > 1 or (bool(single_rank_collectives) and dist.is_available() and dist.is_initialized())
        self.overlap = overlap_allreduce
        if grad_comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('grad_comm_dtype must be torch.float32 or torch.bfloat16')
        self.comm_dtype = grad_comm_dtype
        self._xchg = self._xchg_layout = None
        self._replicas_synced = False

    # -- lr as the reference logs it: scheduler.get_last_lr() after `step_count` scheduler steps
    def get_last_lr(self):
        return self.lr0 * self.mult(self.step_count)

    def _state(self):
        m = self.model.encoder if hasattr(self.model, 'encoder') else self.model
        m._engine()
        if self.m is None or self.m.device != m._pflat.device or self.m.numel() != m._pflat.numel():
            self.m = torch.zeros_like(m._pflat)
            self.v = torch.zeros_like(m._pflat)
            self.norm_out = torch.zeros(2, device=m._pflat.device, dtype=torch.float32)
            self.norm_host = torch.ones(2, dtype=torch.float32).pin_memory()
            self.sumsq_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._flag_event = torch.cuda.Event()
            self._norm_event = torch.cuda.Event()
            self.sumsq = torch.zeros(1, device=m._pflat.device, dtype=torch.float32)
            self.ws = torch.empty(hip.lib().ecgvit_sumsq_workspace(m._pflat.numel()), device=m._pflat.device, dtype=torch.uint8)
            self._replicas_synced = False
        if self.collectives and not self._replicas_synced:
            # data-parallel replicas must start from the same weights whatever each rank's RNG produced: rank 0's flat buffer wins
            ddp.broadcast_flat_(m._pflat, src=0, group=self.pg, single_rank=True)
            if m._wlow is not None:
                m.refresh_low_precision_weights(force=True)
            self._replicas_synced = True

    def _dropout_seed(self, model):
        """one fresh seed per step from the host RNG; the rank is folded in so that replicas seeded identically (as DDP scripts
        do) still drop different units of their different records"""
        if not model._has_dropout:
            return 0
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return (base + 0x3C6EF35F * self.rank) % (2 ** 31 - 1)

    # -- gradient all-reduce (RCCL over xGMI): ddp.GradExchange over the flat gradient buffer
    def _arm_overlap(self, model):
        eng = model._engine()
        if self.collectives:
            if self._xchg is None or self._xchg_layout is not model._layout:
                self._xchg = ddp.GradExchange(model._layout.buckets_in_ready_order(eng.Ly), group=self.pg, overlap=self.overlap,
                                              comm_dtype=self.comm_dtype, single_rank_collectives=True)
                self._xchg_layout = model._layout
            self._xchg.begin(model._gflat)
            eng.on_grads_ready = self._xchg.bucket_ready if self.overlap else None
            # RCCL kernels will hold CUs while the rest of the backward runs: hand the GEMM tiles out in small chunks instead of
            # static per-CU shares (tools/contention.py: 8 held CUs cost a static launch +52 %, a chunked one +9 %) -- an argument of
            # THIS backward pass, not process state
            return 2 if self.overlap else 0
        eng.on_grads_ready = None
        return 0

    def step_masked(self, sample_values, mask_idx):
        """the same fused step for the masked pre-train objective; `self.model` must be a MaskedEcgVit"""
        wrapper, model = self.model, self.model.encoder
        if not model.training:
            raise RuntimeError('train step on a model in eval mode')
        self._state()
        self._raise_if_flagged()
        eng = model._engine()
        seed = self._dropout_seed(model)
        x = sample_values.contiguous().float()
        wrapper.check_mask_indices(mask_idx, x.shape[0])
        if mask_idx.is_cuda:
            idx = mask_idx.to(dtype=torch.int32).contiguous()
        else:
            # host indices (checked on the host above) travel through a pinned staging buffer: a pageable-memory copy would block the host until
            # the device has drained the stream, i.e. once per step.  The buffer is rewritten only after its last copy has left it (an event:
            # normally long past -- with the deferred non-finite check the host may be a step ahead)
            st = getattr(self, '_mask_stage', None)
            if st is None or st[0].shape != mask_idx.shape:
                st = (torch.empty(mask_idx.shape, dtype=torch.int32).pin_memory(), torch.empty(mask_idx.shape, dtype=torch.int32, device=x.device),
                      torch.cuda.Event())
                self._mask_stage = st
            else:
                st[2].synchronize()
            st[0].copy_(mask_idx)
            st[1].copy_(st[0], non_blocking=True)
            st[2].record()
            idx = st[1]
        pred, loss = eng.forward_masked(x, idx, training=True, seed=seed)
        model._fwd_id += 1
        tpw = self._arm_overlap(model)
        try:
            eng.backward_masked(tiles_per_workgroup=tpw)
        finally:
            eng.on_grads_ready = None
        loss, pred = loss.clone(), pred.clone()   # the engine reuses its buffers next step: hand out copies
        self._update(model)
        self.last_loss = loss
        return loss, pred

    def _update(self, model):
        gflat = model._gflat
        if self.collectives:
            self._xchg.finish()
        l = hip.lib()
        st = hip.stream()
        hip.check(l.ecgvit_sumsq(gflat.data_ptr(), gflat.numel(), self.sumsq.data_ptr(), self.ws.data_ptr(), st), 'sumsq')
        # The non-finite decision needs only the sum of squares (the optimiser kernel's test is isfinite(sqrt(sumsq) * |1/world|)): its
        # readback goes out HERE, ahead of the optimiser and the weight-shadow transposes, so that the same-step check below lets the host
        # go on ~0.8 ms before the device has finished the step -- the Python prelude of the next step then runs under those kernels instead
        # of after them (under rocprofv3 the device idled 0.8 ms per step at the step boundary; unprofiled the change is worth 0.2 ms:
        # 6601 -> 6616 records/s, four alternating pairs on one device)
        self.sumsq_host.copy_(self.sumsq, non_blocking=True)
        self._flag_event.record()
        self.step_count += 1
        lr = self.lr0 * self.mult(self.step_count - 1)  # lr in effect for this optimiser step
        hip.check(l.ecgvit_adamw_step(
            model._pflat.data_ptr(), gflat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
            model._wlow.data_ptr() if model._wlow is not None else None, gflat.numel(), self.sumsq.data_ptr(),
            1.0 / self.world, self.max_grad_norm, lr, 0.9, 0.999, 1e-8, self.wd, self.step_count, 1 if self.decoupled else 0,
            self.norm_out.data_ptr(), st), 'adamw_step')
        model.refresh_transposed_weights()   # the optimiser kernel rewrote the bf16 shadows
        # non-blocking readback of (norm, finite flag) as the kernel computed them: only read if the early check fires (the message's norm)
        self.norm_host.copy_(self.norm_out, non_blocking=True)
        self._norm_event.record()
        self._flag_pending = True
        if self.sync_nonfinite:
            self._raise_if_flagged(wait=True)

    def step(self, sample_values, labels):
        model = self.model
        if not model.training:
            raise RuntimeError('train step on a model in eval mode')
        self._state()
        self._raise_if_flagged()
        eng = model._engine()
        seed = self._dropout_seed(model)
        x = sample_values.contiguous().float()
        y = labels.contiguous().float()
        w = None
        if model.loss_weight:
            w = torch.tensor(model.loss_weight, device=y.device, dtype=torch.float32)[y.long()].contiguous()
        logits, _, loss_mean = eng.forward(x, y, w, training=True, seed=seed, want_mean=True)
        model._fwd_id += 1
        B, K = x.shape[0], eng.K
        tpw = self._arm_overlap(model)
        try:
            eng.backward(gscalar=self._one(x.device), gscale=1.0 / (B * K), tiles_per_workgroup=tpw)
        finally:
            eng.on_grads_ready = None
        loss_mean, logits = loss_mean.clone(), logits.clone()   # the engine reuses its buffers next step: hand out copies
        self._update(model)
        self.last_loss = loss_mean
        return loss_mean, logits

    def _one(self, device):
        if getattr(self, '_one_t', None) is None or self._one_t.device != device:
            self._one_t = torch.ones(1, device=device, dtype=torch.float32)
        return self._one_t

    _flag_pending = False

    def _raise_if_flagged(self, wait=False):
        """`clip_grad_norm_(..., error_if_nonfinite=True)` semantics without a per-step host sync: the kernel skips the
        whole update when the norm is non-finite (state stays intact), and the sum of squares it tests is read from pinned memory as
        soon as its copy has landed (`wait=True` blocks for it -- for the copy, issued ahead of the optimiser kernel, not for the step)."""
        if self._flag_pending and (wait or self._flag_event.query()):
            if wait:
                self._flag_event.synchronize()
            self._flag_pending = False
            if not math.isfinite(float(self.sumsq_host[0])):
                self._norm_event.synchronize()
                norm, finite = self.norm_host.tolist()
                if finite == 0.0:
                    raise RuntimeError(f'The total norm for gradients is non-finite ({norm}), so it cannot be clipped.')

    def grad_norm(self):
        """pre-clip global gradient L2 norm of the last step (device sync)"""
        return float(self.norm_out[0].item())

    def finish(self):
        """drain the deferred non-finite check (call after the last step)"""
        self._raise_if_flagged(wait=True)


def clip_grad_norm_(model, max_norm=1.0, error_if_nonfinite=True):
    """`nn.utils.clip_grad_norm_` for the torch-optimizer interop path, on the model's flat gradient buffer (one
    sum-of-squares pass + one scale pass instead of ~150 per-parameter launches). Requires `p.grad` to be the
    engine's gradient views (true after `loss.backward()` when `zero_grad(set_to_none=True)` preceded it)."""
    l = hip.lib()
    g = model._gflat
    for n, p in zip(model._param_names, model._param_list):  # grads that autograd cloned are copied back into the flat buffer
        if p.grad is not None and p.grad.data_ptr() != g.data_ptr() + 4 * model._layout.entries[n][0]:
            model._layout.view(g, n).copy_(p.grad)
            p.grad = model._layout.view(g, n)
    ws = torch.empty(l.ecgvit_sumsq_workspace(g.numel()), device=g.device, dtype=torch.uint8)
    sumsq = torch.empty(1, device=g.device, dtype=torch.float32)
    out = torch.empty(2, device=g.device, dtype=torch.float32)
    hip.check(l.ecgvit_sumsq(g.data_ptr(), g.numel(), sumsq.data_ptr(), ws.data_ptr(), hip.stream()), 'sumsq')
    hip.check(l.ecgvit_clip_scale(g.data_ptr(), g.numel(), sumsq.data_ptr(), float(max_norm), out.data_ptr(), hip.stream()),
              'clip_scale')
    norm, finite = out.tolist()
    if error_if_nonfinite and finite == 0.0:
        raise RuntimeError('The total norm for gradients is non-finite, so it cannot be clipped.')
    return out[0]
