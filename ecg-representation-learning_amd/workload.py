"""
Synthetic workload of the throughput measurement and its algorithmic FLOP count -- product-side helpers used by
`bench.py`; nothing here touches the oracle.

  sample_values  (B, 12, L) f32, i.i.d. N(0, 1): the reference feeds per-lead z-scored signals (preprocess/transform.py:18-35,
                 statistics util/config.json `train-stats`), seed 77 + rank (util/config.json `random-seed` = 77)
  labels         (B, 71) f32 multi-hot, Bernoulli(0.04) (about three positive codes per record; ptb_dataset.py:67-77 builds
                 the same 71-wide multi-hot rows)
"""
import torch


def synthetic_batch(batch, channels=12, length=5000, num_class=71, seed=77, label_p=0.04, device='cpu'):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(batch, channels, length, generator=g, dtype=torch.float32)
    y = (torch.rand(batch, num_class, generator=g) < label_p).to(torch.float32)
    return x.to(device), y.to(device)


def forward_flops_per_record(cfg, num_class=71, cls_token=True):
    """2mnk FLOPs of one forward pass: patch embed 2nCPd + Ly (QKV 6Nd^2 + out 2Nd^2 + FFN 4Ndf + QK^T/PV 4N^2 d) + head 2dK;
    softmax / LayerNorm / GELU arithmetic is not counted"""
    n = cfg.max_signal_length // cfg.patch_size
    N, d, f, ly = n + (1 if cls_token else 0), cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    return 2 * n * cfg.num_channels * cfg.patch_size * d + ly * (8 * N * d * d + 4 * N * d * f + 4 * N * N * d) + 2 * d * num_class


def train_flops_per_record(cfg, num_class=71):
    """train step = 3 x forward (no recomputation credited)"""
    return 3 * forward_flops_per_record(cfg, num_class)


def masked_train_flops_per_record(cfg, mask_ratio=0.5):
    """the masked pre-train step at its own count: trunk over the n patch tokens (no CLS row), patch embed, the pixel head
    Linear(d, C*P) over the m = max(1, int(ratio * n)) masked rows, no classification head; train = 3 x forward"""
    n = cfg.max_signal_length // cfg.patch_size
    d, f, ly, cp = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_channels * cfg.patch_size
    m = max(1, int(mask_ratio * n))
    return 3 * (2 * n * cp * d + ly * (8 * n * d * d + 4 * n * d * f + 4 * n * n * d) + 2 * m * d * cp)
