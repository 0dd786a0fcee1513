"""
Evaluation metrics computed on the device.

Mirror of the reference's `get_accuracy` (`ecg_transformer/util/train.py:12-56`) -- same name, arguments, keys and (quirky)
meanings -- and of `MyTrainer.evaluate` (`ecg_transformer/models/train.py:321-378`).  The reference calls `get_accuracy` on EVERY
train step (`models/train.py:289`): a D2H copy of (B, 71) logits, a host sync, and four sklearn passes.  Here one stream-ordered
`ecgvit_eval_counts` launch leaves 4 + 2*71 exact integers in device memory; they are read (146 x 8 bytes) only when somebody
looks at the numbers, so the train loop stays asynchronous.  The few divisions are done on the host in double precision.

Reference quirks kept, because a drop-in has to log the same numbers (see oracle/metrics_oracle.py for the derivation):
`binary_positive_recall` is tn / (tn + fn) and `binary_negative_recall` is tp / (tp + fp) -- the reference swaps y_true / y_pred in
`classification_report` and then swaps the two names (train.py:46-49).
`per_class_auc` keys: the reference maps class index -> PTB-XL code through its dataset config (`id2code`, train.py:21-22); the
dataset metadata is out of scope here, so pass `id2code` (a list of 71 names) or get integer class indices as keys.
"""
import torch

from . import hip
from .hip import lib, check, ptr, stream


class DeviceCounts:
    """Handle on the integer statistics of one (scores, labels) pair; `.result()` syncs and returns the reference's dict."""

    def __init__(self, counts, B, K, return_auc, id2code):
        self.counts, self.B, self.K, self.return_auc, self.id2code = counts, B, K, return_auc, id2code

    def result(self):
        c = self.counts.cpu().tolist()   # the only D2H transfer (and sync) of the metric path
        B, K = self.B, self.K
        tp, tn, fp, fn = c[:4]
        recalls = [r for r in ((tn / (tn + fp)) if tn + fp else None, (tp / (tp + fn)) if tp + fn else None) if r is not None]
        macro, per_class = None, None
        if self.return_auc:
            pos, pairs = c[4:4 + K], c[4 + K:]
            valid = [k for k in range(K) if 0 < pos[k] < B]   # classes with both label values (train.py:29)
            if valid:
                per_class = {(self.id2code[k] if self.id2code is not None else k): pairs[k] / (2.0 * pos[k] * (B - pos[k])) for k in valid}
                vals = list(per_class.values())
                macro = sum(vals) / len(vals)
        return dict(
            binary_accuracy=(tp + tn) / (B * K),
            weighted_binary_accuracy=sum(recalls) / len(recalls),
            binary_negative_recall=tp / (tp + fp) if tp + fp else 0.0,
            binary_positive_recall=tn / (tn + fn) if tn + fn else 0.0,
            macro_auc=macro, per_class_auc=per_class)


def eval_counts(scores, labels, from_logits=False, return_auc=True, id2code=None) -> DeviceCounts:
    """Launch the counting kernels (asynchronous). scores / labels: (B, K) f32 device tensors."""
    if not (scores.is_cuda and labels.is_cuda):
        raise RuntimeError('metrics run on the device (no CPU fallback exists): pass device tensors')
    assert scores.dim() == 2 and scores.shape == labels.shape and scores.dtype == torch.float32 and labels.dtype == torch.float32
    assert scores.stride(1) == 1 and labels.stride(1) == 1
    B, K = scores.shape
    counts = torch.empty(4 + 2 * K, dtype=torch.int64, device=scores.device)
    check(lib().ecgvit_eval_counts(ptr(scores), scores.stride(0), ptr(labels), labels.stride(0), B, K, int(from_logits), int(return_auc),
                                   ptr(counts), stream()), 'eval_counts')
    h = DeviceCounts(counts, B, K, return_auc, id2code)
    h._keep = (scores, labels)
    return h


def get_accuracy(preds, labels, return_auc=True, id2code=None):
    """Drop-in for the reference's `get_accuracy(preds, labels, return_auc)`: `preds` are per-class probabilities."""
    return eval_counts(preds, labels, from_logits=False, return_auc=return_auc, id2code=id2code).result()


class HipEvaluator:
    """`MyTrainer.evaluate` (models/train.py:321-378) with everything kept on the device: logits of the whole eval set are written
    into one (n_eval, K) buffer, the loss is the mean of the per-batch mean losses (train.py:357-358), and the metrics come from one
    `ecgvit_eval_counts` over the whole set. With torch.distributed initialised and `gather=True` each rank evaluates its shard of the
    records and the (logits, labels) shards are all-gathered (RCCL) before counting, so every rank reports whole-set AUROC."""

    def __init__(self, model, eval_batch_size=64, id2code=None, gather=False, process_group=None):
        self.model, self.bsz, self.id2code, self.gather, self.pg = model, int(eval_batch_size), id2code, gather, process_group

    def evaluate(self, sample_values, labels, return_predictions=False):
        """sample_values: (n, C, L) f32 device tensor (this rank's shard), labels: (n, K) f32"""
        model = self.model
        training = model.training
        model.eval()
        n, K = labels.shape
        logits = torch.empty(n, K, dtype=torch.float32, device=labels.device)
        losses = []
        with torch.no_grad():
            for s in range(0, n, self.bsz):
                out = model(sample_values=sample_values[s:s + self.bsz], labels=labels[s:s + self.bsz])
                logits[s:s + self.bsz] = out.logits
                losses.append(out.loss.detach().reshape(()))
        loss = torch.stack(losses).mean()
        lb = labels
        if self.gather and torch.distributed.is_available() and torch.distributed.is_initialized():
            import torch.distributed as dist
            ws = dist.get_world_size(self.pg)
            sizes = [torch.zeros(1, dtype=torch.int64, device=labels.device) for _ in range(ws)]
            dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=labels.device), group=self.pg)
            sizes = [int(v) for v in sizes]
            mx = max(sizes)

            def gather(t):
                pad = torch.zeros(mx, K, dtype=t.dtype, device=t.device)
                pad[:n] = t
                parts = [torch.empty_like(pad) for _ in range(ws)]
                dist.all_gather(parts, pad, group=self.pg)
                return torch.cat([p[:m] for p, m in zip(parts, sizes)], dim=0)
            logits, lb = gather(logits), gather(labels)
            wsum = loss * len(losses)
            cnt = torch.tensor([float(len(losses))], device=labels.device)
            dist.all_reduce(wsum, group=self.pg)
            dist.all_reduce(cnt, group=self.pg)
            loss = wsum / cnt[0]
        m = eval_counts(logits, lb.contiguous(), from_logits=True, return_auc=True, id2code=self.id2code).result()
        d = {'eval/loss': float(loss), **{f'eval/{k}': v for k, v in m.items()}}
        if training:
            model.train()
        return dict(metrics=d, predictions=dict(labels=lb, logits=logits)) if return_predictions else d
