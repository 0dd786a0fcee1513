"""
CPU fp32 restatement of the reference hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Two layers, as in the reference:

  OracleViT        the arithmetic the reference delegates to third-party
                   `vit-pytorch==0.33.2` (reference call sites: `models/ecg_vit.py:12`,
                   `:102-116` ctor kwargs, `:141` forward, `:176-180` Recorder, `:277`
                   `to_patch_embedding`).  Package absent -> published algorithm restated;
                   **parity unpinned** for this class.
  OracleEcgVit     the reference's own adapter, `models/ecg_vit.py:95-149`.

plus the train-step body of `models/train.py:268-283` (`oracle_train_step`) and the
SimMIM-style masked objective the reference does not have (`OracleMaskedEcgVit`,
**parity unpinned**).

Module/parameter names reproduce the reference checkpoint's `state_dict` key layout
(strict load at `models/ecg_vit.py:159`):
  vit.pos_embedding, vit.cls_token, vit.to_patch_embedding.1.{weight,bias},
  vit.transformer.layers.{i}.0.norm.*, ...0.fn.to_qkv.weight, ...0.fn.to_out.0.*,
  vit.transformer.layers.{i}.1.norm.*, ...1.fn.net.0.*, ...1.fn.net.3.*,
  vit.mlp_head.0.*, vit.mlp_head.1.*
"""
from collections import namedtuple
import math

import numpy as np
import torch
from torch import nn

# reference: ecg_transformer/util/models.py:3
ModelOutput = namedtuple('ModelOutput', ['loss', 'logits'])


# --------------------------------------------------------------------------------------
# integer part: patch gather (bit-exact pin)
# --------------------------------------------------------------------------------------
def patch_gather_np(x: np.ndarray, patch: int) -> np.ndarray:
    """
    `Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p1=1, p2=P)` applied to the
    fake-2D image `(B, C, 1, L)` the reference builds at `models/ecg_vit.py:141`.

    out[b, p, j*C + c] = x[b, c, p*P + j]   (sample-major, lead-minor inside a patch)
    """
    b, c, l = x.shape
    assert l % patch == 0, 'Image dimensions must be divisible by the patch size.'
    n = l // patch
    out = np.empty((b, n, patch * c), dtype=x.dtype)
    for j in range(patch):
        for ch in range(c):
            out[:, :, j * c + ch] = x[:, ch, j::patch][:, :n]
    return out


def patch_gather(x: torch.Tensor, patch: int) -> torch.Tensor:
    b, c, l = x.shape
    assert l % patch == 0, 'Image dimensions must be divisible by the patch size.'
    n = l // patch
    # (b, c, n, P) -> (b, n, P, c) -> (b, n, P*c)
    return x.reshape(b, c, n, patch).permute(0, 2, 3, 1).reshape(b, n, patch * c)


class _PatchRearrange(nn.Module):
    """index 0 of `to_patch_embedding` (parameter-free, so Linear gets key `.1.`)"""

    def __init__(self, patch):
        super().__init__()
        self.patch = patch

    def forward(self, img):  # img: (B, C, 1, L)
        return patch_gather(img.squeeze(-2), self.patch)


# --------------------------------------------------------------------------------------
# third-party arithmetic, restated  (vit-pytorch 0.33.2, parity unpinned)
# --------------------------------------------------------------------------------------
class InjectedDropout(nn.Dropout):
    """`nn.Dropout` (the module vit-pytorch places at all five sites: embedding, attention probabilities, `to_out`, FFN hidden,
    FFN output -- reference `models/ecg_vit.py:113-114` sets their p) whose multiplier can be INJECTED: with `mult` set to the
    tensor of 0 / 1/(1-p) factors another implementation drew for this site, forward is `x * mult` in train AND eval mode, so the
    oracle reproduces that implementation's dropout realisation exactly and a p > 0 step becomes comparable number for number.
    `mult is None` (default): plain `nn.Dropout` -- torch's own generator, as in the reference."""
    mult = None

    def forward(self, x):
        if self.mult is None:
            return super().forward(x)
        assert self.mult.numel() == x.numel(), (tuple(self.mult.shape), tuple(x.shape))
        return x * self.mult.reshape(x.shape).to(x.dtype)


def dropout_sites(vit):
    """the five kinds of dropout site of an `OracleViT`, in the order the forward reaches them:
    {'emb': module, 'layers': [{'probs', 'out', 'ffn', 'down'}: module per layer]}"""
    layers = []
    for attn, ff in vit.transformer.layers:
        layers.append(dict(probs=attn.fn.dropout, out=attn.fn.to_out[1], ffn=ff.fn.net[2], down=ff.fn.net[4]))
    return dict(emb=vit.dropout, layers=layers)


def inject_dropout(vit, masks):
    """masks: {'emb': (B, T, d) | None, 'layers': [{'probs': (B, h, T, T), 'out': (B, T, d), 'ffn': (B, T, f), 'down': (B, T, d)}]}
    multipliers (0 or 1/(1-p)); None entries / `masks is None` restore torch's own dropout at that site"""
    sites = dropout_sites(vit)
    sites['emb'].mult = None if masks is None else masks.get('emb')
    for i, s in enumerate(sites['layers']):
        for k, mod in s.items():
            mod.mult = None if masks is None else masks['layers'][i].get(k)


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)  # eps 1e-5, biased variance, affine
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)  # Recorder hooks this module's output (ecg_vit.py:176)
        self.dropout = InjectedDropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        project_out = not (heads == 1 and dim_head == dim)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), InjectedDropout(dropout)) if project_out else nn.Identity()

    def forward(self, x):
        b, n, _ = x.shape
        h, dh = self.heads, self.dim_head
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)  # rows of to_qkv.weight ordered [q; k; v], head-major
        q, k, v = (t.reshape(b, n, h, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        dots = torch.matmul(q, k.transpose(-1, -2)) * self.scale  # scale AFTER QK^T
        attn = self.dropout(self.attend(dots))
        out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, h * dh)
        return self.to_out(out)


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden, dropout):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, hidden), nn.GELU(), InjectedDropout(dropout), nn.Linear(hidden, dim), InjectedDropout(dropout)
        )

    def forward(self, x):
        return self.net(x)


class _Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([
                _PreNorm(dim, _Attention(dim, heads, dim_head, dropout)),
                _PreNorm(dim, _FeedForward(dim, mlp_dim, dropout)),
            ]) for _ in range(depth)
        ])

    def forward(self, x):
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x  # no final LN in the trunk; it sits in the head


class OracleViT(nn.Module):
    """Constructor signature = the kwargs the reference passes at `models/ecg_vit.py:102-115`."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls',
                 channels=3, dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        ih, iw = image_size if isinstance(image_size, tuple) else (image_size, image_size)
        ph, pw = patch_size if isinstance(patch_size, tuple) else (patch_size, patch_size)
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        assert ih == 1 and ph == 1, 'oracle covers the 1-D (height 1) use the reference makes of ViT'
        assert pool in {'cls', 'mean'}
        n_patch = (ih // ph) * (iw // pw)
        self.to_patch_embedding = nn.Sequential(_PatchRearrange(pw), nn.Linear(channels * ph * pw, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n_patch + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = InjectedDropout(emb_dropout)
        self.transformer = _Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def trunk(self, img):
        x = self.to_patch_embedding(img)
        b, n, _ = x.shape
        x = torch.cat((self.cls_token.expand(b, -1, -1), x), dim=1)
        x = x + self.pos_embedding[:, :(n + 1)]
        x = self.dropout(x)
        return self.transformer(x)

    def forward(self, img):
        x = self.trunk(img)
        x = x.mean(dim=1) if self.pool == 'mean' else x[:, 0]
        return self.mlp_head(self.to_latent(x))


# --------------------------------------------------------------------------------------
# the reference's own adapter  (models/ecg_vit.py:95-149)
# --------------------------------------------------------------------------------------
class OracleEcgVit(nn.Module):
    """
    Takes any object with the `EcgVitConfig` fields (ecg_vit.py:26-54).
    NB (ecg_vit.py:105,113-114): ctor `num_class` is used, `config.num_class` ignored;
    ViT `dropout` <- hidden_dropout_prob (attention probs + FFN + to_out),
    ViT `emb_dropout` <- attention_probs_dropout_prob.
    """

    def __init__(self, num_class=71, config=None, loss_reduction='mean'):
        super().__init__()
        c = config
        assert c.hidden_size % c.num_attention_heads == 0  # ecg_vit.py:99
        self.config = c
        self.vit = OracleViT(
            image_size=(1, c.max_signal_length), patch_size=(1, c.patch_size), num_classes=num_class,
            dim=c.hidden_size, depth=c.num_hidden_layers, heads=c.num_attention_heads,
            mlp_dim=c.intermediate_size, pool='cls', channels=c.num_channels,
            dim_head=c.hidden_size // c.num_attention_heads,
            dropout=c.hidden_dropout_prob, emb_dropout=c.attention_probs_dropout_prob)
        self._loss_reduction = loss_reduction
        self.loss_weight = None

    @property
    def loss_reduction(self):
        return self._loss_reduction

    @loss_reduction.setter
    def loss_reduction(self, r):
        self._loss_reduction = r

    def forward(self, sample_values, labels=None):
        logits = self.vit(sample_values.unsqueeze(-2))  # ecg_vit.py:141
        loss = None
        if labels is not None:
            weight = None
            if self.loss_weight:  # ecg_vit.py:144-147: per-element weight looked up by label value
                weight = torch.tensor(self.loss_weight, device=labels.device)[labels.long()]
            loss = bce_with_logits(logits, labels, weight=weight, reduction=self._loss_reduction)
        return ModelOutput(loss=loss, logits=logits)


def bce_with_logits(z, y, weight=None, reduction='mean'):
    """nn.BCEWithLogitsLoss (ecg_vit.py:118,148): l = max(z,0) - z*y + log1p(exp(-|z|))"""
    l = z.clamp(min=0) - z * y + torch.log1p(torch.exp(-z.abs()))
    if weight is not None:
        l = l * weight
    if reduction == 'mean':
        return l.mean()
    if reduction == 'sum':
        return l.sum()
    return l


# --------------------------------------------------------------------------------------
# train step  (models/train.py:241-252 setup, :268-283 step body)
# --------------------------------------------------------------------------------------
def lr_lambda(schedule: str, n_warmup: int, n_step: int):
    """
    HF `get_constant_schedule_with_warmup` / `get_cosine_schedule_with_warmup` (num_cycles 0.5),
    selected at train.py:245-252 with num_warmup_steps = round(n_step * warmup_ratio).
    Returns f(step) -> lr multiplier, step = number of scheduler.step() calls so far.
    """
    def f(step):
        if step < n_warmup:
            return float(step) / float(max(1, n_warmup))
        if schedule == 'constant':
            return 1.0
        prog = float(step - n_warmup) / float(max(1, n_step - n_warmup))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))
    return f


def clip_grad_norm(params, max_norm=1.0, error_if_nonfinite=True):
    """torch.nn.utils.clip_grad_norm_ semantics (train.py:281): global L2 over all grads,
    coef = max_norm / (norm + 1e-6) clamped to 1, RuntimeError on non-finite norm."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    if error_if_nonfinite and not bool(torch.isfinite(total)):
        raise RuntimeError('The total norm for gradients is non-finite, so it cannot be clipped.')
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(params, state, lr, weight_decay, step, betas=(0.9, 0.999), eps=1e-8, decoupled=True):
    """torch.optim.AdamW (decoupled=True) / Adam (False) single step, torch-default hyper-parameters."""
    b1, b2 = betas
    for p in params:
        if p.grad is None:
            continue
        g = p.grad
        st = state.setdefault(p, dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
        if decoupled:
            p.data.mul_(1 - lr * weight_decay)
        elif weight_decay != 0:
            g = g + weight_decay * p.data
        st['m'].mul_(b1).add_(g, alpha=1 - b1)
        st['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(eps)
        p.data.addcdiv_(st['m'], denom, value=-lr / bc1)


class OracleTrainer:
    """
    zero_grad -> forward -> backward -> clip_grad_norm_(1.0) -> optimizer.step -> scheduler.step
    (train.py:271-283), with torch.optim.AdamW/Adam + HF LambdaLR exactly as the reference wires them.
    """

    def __init__(self, model, learning_rate=3e-4, weight_decay=1e-2, optimizer='AdamW',
                 schedule='cosine', warmup_ratio=0.05, n_step=1000):
        self.model = model
        cls = torch.optim.AdamW if optimizer == 'AdamW' else torch.optim.Adam
        self.optimizer = cls(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lr_lambda(schedule, round(n_step * warmup_ratio), n_step))
        self.last_grad_norm = None

    def step(self, sample_values, labels):
        self.optimizer.zero_grad()
        out = self.model(sample_values=sample_values, labels=labels)
        out.loss.backward()
        self.last_grad_norm = nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=1.0, error_if_nonfinite=True)
        self.optimizer.step()
        self.scheduler.step()
        return out


# --------------------------------------------------------------------------------------
# masked pre-train objective -- ABSENT from the reference; build's own definition
# --------------------------------------------------------------------------------------
class OracleMaskedEcgVit(nn.Module):
    """
    SimMIM-style: tokens = Linear(patches) ; masked tokens <- mask_token ; + pos[1:n+1] ;
    trunk on n tokens (no CLS) ; gather masked rows -> Linear(d, C*P) ; L1 vs raw patches / (m * C*P).
    Shares `encoder.vit` weights.  **parity unpinned** (nothing in the reference to compare to).
    """

    def __init__(self, encoder: OracleEcgVit):
        super().__init__()
        self.encoder = encoder
        c = encoder.config
        d, cp = c.hidden_size, c.num_channels * c.patch_size
        self.mask_token = nn.Parameter(torch.randn(d))
        self.to_pixels = nn.Linear(d, cp)

    def forward(self, sample_values, mask_idx):
        """mask_idx: (B, m) int, distinct patch indices per record."""
        vit, c = self.encoder.vit, self.encoder.config
        patches = patch_gather(sample_values, c.patch_size)  # (B, n, C*P)
        b, n, _ = patches.shape
        tok = vit.to_patch_embedding[1](patches)
        is_masked = torch.zeros(b, n, dtype=torch.bool, device=tok.device)
        is_masked.scatter_(1, mask_idx.long(), True)
        tok = torch.where(is_masked.unsqueeze(-1), self.mask_token.expand(b, n, -1), tok)
        tok = tok + vit.pos_embedding[:, 1:(n + 1)]
        tok = vit.dropout(tok)
        enc = vit.transformer(tok)
        bi = torch.arange(b, device=tok.device).unsqueeze(-1)
        pred = self.to_pixels(enc[bi, mask_idx.long()])  # (B, m, C*P)
        target = patches[bi, mask_idx.long()]
        loss = (pred - target).abs().mean()
        return ModelOutput(loss=loss, logits=pred)


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench (synthetic inputs of SURVEY §8d)
# --------------------------------------------------------------------------------------
def synthetic_batch(batch, channels=12, length=5000, num_class=71, seed=77, label_p=0.04, device='cpu'):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(batch, channels, length, generator=g, dtype=torch.float32)
    y = (torch.rand(batch, num_class, generator=g) < label_p).to(torch.float32)
    return x.to(device), y.to(device)


def train_flops_per_record(cfg, num_class=71):
    """SURVEY §8d algorithmic FLOPs: train = 3 x forward, forward = 2nCPd + Ly(24Nd^2... general f) + 2dK"""
    n = cfg.max_signal_length // cfg.patch_size
    N, d, f, ly = n + 1, cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    fwd = 2 * n * cfg.num_channels * cfg.patch_size * d + ly * (8 * N * d * d + 4 * N * d * f + 4 * N * N * d) + 2 * d * num_class
    return 3 * fwd
