"""
Host-side mirror of the reference model interface for the hot path, backed by the gfx950 kernels.

Same names, arguments, defaults and error behaviour as `ecg_transformer/models/ecg_vit.py`:
  EcgVitConfig (:26-92), EcgVit (:95-149), ModelOutput (`ecg_transformer/util/models.py:3`);
`state_dict()` keys/shapes are those of the reference checkpoint (strict load at ecg_vit.py:159), so
reference `.pt` files load here and vice versa.

What differs is only what executes: `EcgVit.forward` launches hand-written HIP kernels through the C-ABI
(`include/ecgvit_hip.h`) and `loss.backward()` runs the hand-written backward; the `nn.Module` tree under
`.vit` is a PARAMETER CONTAINER with the vit-pytorch 0.33.2 attribute layout (`to_patch_embedding`,
`transformer.layers[i][0].fn.to_qkv`, ..., `mlp_head`), not an eager implementation.  There is no CPU path:
calling the model with host tensors, or without the built library, raises.
"""
import re
from collections import namedtuple

import numpy as np
import torch
from torch import nn

try:  # the reference subclasses HF PretrainedConfig (to_dict / save_pretrained / repr); present in this image
    from transformers import PretrainedConfig as _ConfigBase
except Exception:  # pragma: no cover - minimal stand-in when transformers is absent
    class _ConfigBase:
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                setattr(self, k, v)

        def to_dict(self):
            return dict(self.__dict__)

from .check_args import ca
from .engine import VitEngine, ParamLayout
from . import hip

ModelOutput = namedtuple('ModelOutput', ['loss', 'logits'])  # reference util/models.py:3


def log_dict_p(d):
    """reference util/util.py:326-330 (`log_dict(with_color=False, sep='=')`) for the str/int values `meta_str` holds"""
    return '{' + ', '.join(f'{k}={v}' for k, v in d.items()) + '}'


class EcgVitConfig(_ConfigBase):
    """Field names / defaults: reference ecg_vit.py:29-54."""
    pattern_model_name = re.compile(r'^(?P<name>\S+)-(?P<size>\S+)$')

    def __init__(
            self,
            max_signal_length: int = 2560,
            patch_size: int = 64,
            num_channels: int = 12,
            hidden_size: int = 512,
            num_hidden_layers: int = 8,
            num_attention_heads: int = 8,
            intermediate_size: int = 2048,
            hidden_dropout_prob: float = 0.1,
            attention_probs_dropout_prob: float = 0.1,
            num_class: int = 71,
            **kwargs
    ):
        self.max_signal_length = max_signal_length
        self.patch_size = patch_size
        self.num_channels = num_channels
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.num_class = num_class
        super().__init__(**kwargs)
        self.size = None

    _SIZES = dict(  # reference ecg_vit.py:67-91
        debug=(64, 4, 4, 256), tiny=(256, 4, 4, 1024), small=(512, 8, 8, 2048), base=(768, 12, 12, 3072),
        large=(1024, 24, 16, 4096))

    @classmethod
    def from_defined(cls, model_name):
        ca(model_name=model_name)
        conf = cls()
        m = cls.pattern_model_name.match(model_name)
        nm, size = m.group('name'), m.group('size')
        conf.size = size
        assert nm == 'ecg-vit'
        conf.hidden_size, conf.num_hidden_layers, conf.num_attention_heads, conf.intermediate_size = cls._SIZES[size]
        return conf


# ----------------------------------------------------------------------------------------------------------
# parameter container with the vit-pytorch 0.33.2 attribute / state_dict layout
# ----------------------------------------------------------------------------------------------------------
def _container_only(self, *a, **k):
    raise RuntimeError('this sub-module is a parameter container; it executes only inside EcgVit.forward (HIP engine)')


class _Rearrange(nn.Module):  # index 0 of to_patch_embedding: parameter-free, keeps the Linear at key `.1.`
    forward = _container_only


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn
    forward = _container_only


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))
    forward = _container_only


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden, dropout):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden, dim),
                                 nn.Dropout(dropout))
    forward = _container_only


class _Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, _Attention(dim, heads, dim_head, dropout)),
                           _PreNorm(dim, _FeedForward(dim, mlp_dim, dropout))]) for _ in range(depth)])
    forward = _container_only


class _PatchEmbedding(nn.Sequential):
    """`model.vit.to_patch_embedding(x.unsqueeze(-2))` is called on its own by the reference (ecg_vit.py:277)."""

    def forward(self, img):
        return self._owner().patch_embed(img.squeeze(-2))


class HipViT(nn.Module):
    """Same constructor kwargs as the call at reference ecg_vit.py:102-115."""

    def __init__(self, *, image_size, patch_size, num_classes, dim, depth, heads, mlp_dim, pool='cls', channels=3,
                 dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        (ih, iw), (ph, pw) = image_size, patch_size
        assert ih % ph == 0 and iw % pw == 0, 'Image dimensions must be divisible by the patch size.'
        assert pool == 'cls' and ih == 1 and ph == 1
        n_patch = iw // pw
        self.to_patch_embedding = _PatchEmbedding(_Rearrange(), nn.Linear(channels * pw, dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, n_patch + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = _Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.pool = pool
        self.to_latent = nn.Identity()
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
    forward = _container_only


# ----------------------------------------------------------------------------------------------------------
class _EcgVitFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward and backward are the engine's kernel schedules."""

    @staticmethod
    def forward(ctx, model, x, labels, weight, reduction, *params):
        eng = model._engine()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (model.training and model._has_dropout) else 0
        logits, loss_elem, loss_mean = eng.forward(x, labels, weight, training=model.training, seed=seed,
                                                   want_mean=(reduction == 'mean'))
        model._fwd_id += 1
        ctx.model, ctx.fwd_id, ctx.reduction, ctx.has_labels = model, model._fwd_id, reduction, labels is not None
        ctx.set_materialize_grads(False)
        out_logits = logits.clone()
        if labels is None:
            loss = logits.new_zeros(())
        elif reduction == 'mean':
            loss = loss_mean.clone().reshape(())
        elif reduction == 'sum':
            raise NotImplementedError("loss_reduction 'sum' is not part of the reference's contract ('mean' | 'none')")
        else:
            loss = loss_elem.clone()
        return loss, out_logits

    @staticmethod
    def backward(ctx, gloss, glogits):
        model = ctx.model
        if ctx.fwd_id != model._fwd_id:
            raise RuntimeError('EcgVit: a later forward overwrote the activations this backward needs '
                               '(one live graph per model; call backward before the next forward)')
        eng = model._engine()
        B, K = eng.saved['B'], eng.K
        if (gloss is None or not ctx.has_labels) and glogits is None:
            return (None,) * (5 + len(model._own_list))
        keep, aliased = _grads_living_in_flat_buffer(model, model._own_names, model._own_list)
        if gloss is not None and ctx.has_labels:
            if glogits is not None:
                raise NotImplementedError('gradients through both loss and logits of one forward')
            if ctx.reduction == 'mean':
                eng.backward(gscalar=gloss.contiguous().float(), gscale=1.0 / (B * K))
            else:
                eng.backward(gelem=gloss.contiguous().float(), gscale=1.0)
        else:
            eng.backward(glogits=glogits.contiguous().float())
        return (None, None, None, None, None) + _grads_out(model, model._own_names, keep, aliased)


def _grads_living_in_flat_buffer(model, names, params):
    """Parameters whose `.grad` IS a view of the engine's flat gradient buffer (AccumulateGrad adopted the views a previous backward
    returned).  The engine overwrites that buffer, so an accumulating backward (`zero_grad(set_to_none=False)`, gradient accumulation
    over micro-batches) would otherwise compute `p.grad += p.grad`-of-the-new-values: keep a copy of the old contents to add back.
    Returns (copy of the flat buffer or None, set of aliased names)."""
    g = model._gflat
    aliased = {n for n, p in zip(names, params)
               if p.grad is not None and p.grad.data_ptr() == g.data_ptr() + 4 * model._layout.entries[n][0]}
    return (g.clone() if aliased else None), aliased


def _grads_out(model, names, keep, aliased):
    """gradient tuple for autograd: aliased parameters were accumulated in place (old + new) and return None; the others return their
    view of the flat buffer (adopted as `.grad` when it was None, added to a foreign `.grad` otherwise)"""
    out = []
    for n in names:
        v = model._layout.view(model._gflat, n)
        if n in aliased:
            v.add_(model._layout.view(keep, n))
            out.append(None)
        else:
            out.append(v)
    return tuple(out)


class EcgVit(nn.Module):
    """
    reference ecg_vit.py:95-149.  Extra (keyword-only in spirit) argument `compute_dtype`:
      torch.float32  -- parity path (exact-f32 MFMA; reproduces the reference's CPU numbers to ~1e-6)
      torch.bfloat16 -- throughput path (bf16 MFMA GEMMs + fused attention, f32 accumulate / statistics / master weights)
    `fp8_linear=True` (with bf16): the block Linears' forward and input-gradient products run on the CDNA4 fp8 MFMA with per-tensor
    scaled e4m3 / e5m2 operands (BASELINE.json configs[4]); master weights, optimiser, weight gradients, attention stay as in bf16.
    `saved_ffn_e4m3` (bf16 only): the tensor the FFN backward needs, gelu'(pre) x dropout multiplier, is kept between forward and backward as
    e4m3 bytes (None = default: on, over >= 2048 token rows; relative error <= 2^-4 per element, unbiased, forward values unaffected, whole-gradient
    cosine to the bf16 form > 0.999) -- False keeps it in bf16 (+1.2 % step time at base).
    """

    def __init__(self, num_class: int = 71, config=None, loss_reduction: str = 'mean', compute_dtype=torch.float32, fp8_linear=False, saved_ffn_e4m3=None):
        super().__init__()
        self.saved_ffn_e4m3 = saved_ffn_e4m3
        self.fp8_linear = bool(fp8_linear)   # bf16 path with e4m3 / e5m2 operands in the block Linears' forward and input-gradient products
        config = config if config is not None else EcgVitConfig()
        hd_sz, n_head = config.hidden_size, config.num_attention_heads
        assert hd_sz % n_head == 0
        dim_head = hd_sz // n_head
        self.config = config
        self.num_class = num_class
        self.vit = HipViT(
            image_size=(1, config.max_signal_length), patch_size=(1, config.patch_size), num_classes=num_class,
            dim=config.hidden_size, depth=config.num_hidden_layers, heads=config.num_attention_heads,
            mlp_dim=config.intermediate_size, pool='cls', channels=config.num_channels, dim_head=dim_head,
            dropout=config.hidden_dropout_prob, emb_dropout=config.attention_probs_dropout_prob)
        object.__setattr__(self.vit.to_patch_embedding, '_owner', lambda: self)  # plain attribute: no module cycle
        self._loss_reduction = loss_reduction
        self.loss_fn = nn.BCEWithLogitsLoss(reduction=loss_reduction)  # attribute kept for callers that poke at it
        self.loss_weight = None

        C, L = config.num_channels, config.max_signal_length
        cls_nm = self.__class__.__qualname__
        n_pch, n_l, n_h = L // config.patch_size, config.num_hidden_layers, config.num_attention_heads
        self.meta = {'name': cls_nm, 'input shape': f'{C} x {L}', '#patch': n_pch, '#layer': n_l, '#head': n_h}
        self.meta_str = log_dict_p({'nm': cls_nm, 'in-sp': f'{C}x{L}', '#p': n_pch, '#l': n_l, '#h': n_h})

        self.compute_dtype = compute_dtype
        self._has_dropout = config.hidden_dropout_prob > 0 or config.attention_probs_dropout_prob > 0
        self._fwd_id = 0
        self._eng = None
        self._pflat = self._gflat = self._wlow = self._wlow_t = self._tr_table = None
        self._wlow_version = -1
        self._own_names = [n for n, _ in self.named_parameters()]
        self._own_list = [p for _, p in self.named_parameters()]
        self._param_names, self._param_list = list(self._own_names), list(self._own_list)  # + attached extras (pre-train head)
        self._layout = ParamLayout([(n, tuple(p.shape)) for n, p in zip(self._param_names, self._param_list)])
        self._flatten()

    def _attach_extra(self, named_params):
        """append parameters owned by a wrapper (the masked pre-train head) to the flat HBM layout"""
        self._param_names = list(self._own_names) + [n for n, _ in named_params]
        self._param_list = list(self._own_list) + [p for _, p in named_params]
        self._layout = ParamLayout([(n, tuple(p.shape)) for n, p in zip(self._param_names, self._param_list)])
        self._flatten()

    # ------------------------------------------------------------------ reference surface
    def to_str(self):
        return f'{self.__class__.__qualname__}, {self.config.size}'

    @property
    def loss_reduction(self):
        return self._loss_reduction

    @loss_reduction.setter
    def loss_reduction(self, r):
        self.loss_fn.reduction = self._loss_reduction = r

    def forward(self, sample_values: torch.FloatTensor, labels: torch.LongTensor = None):
        if not sample_values.is_cuda:
            raise RuntimeError('EcgVit (HIP) runs on an MI355X device only: move the model and inputs to "cuda" '
                               '(there is deliberately no CPU fallback)')
        x = sample_values.contiguous().float()
        y = w = None
        if labels is not None:
            y = labels.contiguous().float()
            if self.loss_weight:  # reference :144-147: per-element weight looked up by the label value
                w = torch.tensor(self.loss_weight, device=y.device, dtype=torch.float32)[y.long()].contiguous()
        loss, logits = _EcgVitFunction.apply(self, x, y, w, self._loss_reduction, *self._own_list)
        return ModelOutput(loss=loss if labels is not None else None, logits=logits)

    # ------------------------------------------------------------------ flat HBM layout of the parameters
    def _flatten(self):
        """(Re)pack all parameters into one flat f32 buffer on their current device; params become views of it."""
        dev = self._param_list[0].device
        flat = torch.zeros(self._layout.total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in zip(self._param_names, self._param_list):
                v = self._layout.view(flat, n)
                v.copy_(p.data)
                p.data = v
        self._pflat = flat
        self._gflat = torch.zeros_like(flat)
        self._wlow = self._wlow_t = self._tr_table = None
        self._eng = None

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._pflat is not None:
            self._flatten()
        return out

    def _is_flat(self):
        base = self._pflat.data_ptr()
        for n, p in zip(self._param_names, self._param_list):
            if p.data_ptr() != base + 4 * self._layout.entries[n][0] or p.dtype != torch.float32:
                return False
        return True

    def _engine(self):
        if not torch.cuda.is_available():
            raise RuntimeError('no HIP device')
        if not self._is_flat():
            self._flatten()
        if self._eng is None or self._eng.dtype != self.compute_dtype:
            c = self.config
            self._eng = VitEngine(C=c.num_channels, L=c.max_signal_length, P=c.patch_size, d=c.hidden_size,
                                  h=c.num_attention_heads, f=c.intermediate_size, Ly=c.num_hidden_layers, K=self.num_class,
                                  p_hidden=c.hidden_dropout_prob, p_emb=c.attention_probs_dropout_prob,
                                  dtype=self.compute_dtype, layout=self._layout, fp8_linear=self.fp8_linear, saved_ffn_e4m3=self.saved_ffn_e4m3)
            self._wlow_t = self._tr_table = None
            if self.compute_dtype == torch.bfloat16:
                self._wlow = torch.empty(self._layout.total, dtype=torch.bfloat16, device=self._pflat.device)
                self._wlow_version = -1
                self._tr_table, self._tr_nmat, self._tr_tiles = self._eng.transposed_weight_table(self._pflat.device)
                if self._tr_table is not None:
                    self._wlow_t = torch.zeros(self._layout.total, dtype=torch.bfloat16, device=self._pflat.device)
            self._eng.bind(self._pflat, self._gflat, self._wlow, self._wlow_t)
            self._eng.input_transform = getattr(self, '_input_transform', None)
        if self.compute_dtype == torch.bfloat16:
            self.refresh_low_precision_weights()
        return self._eng

    def refresh_low_precision_weights(self, force=False):
        """bf16 shadow of the master weights; refreshed when any parameter was modified in place since the last cast
        (the fused optimiser writes it itself and bumps nothing)."""
        ver = sum(p._version for p in self._param_list) + self._pflat._version
        if force or ver != self._wlow_version:
            hip.check(hip.lib().ecgvit_cast_f32_to_bf16(self._pflat.data_ptr(), self._wlow.data_ptr(), self._layout.total,
                                                        hip.stream()), 'cast_f32_to_bf16')
            self.refresh_transposed_weights()
            self._wlow_version = ver

    def refresh_transposed_weights(self):
        """W^T shadows of the block Linears (input-gradient GEMMs then run on the forward kernel); call after `_wlow` changed"""
        if getattr(self, '_wlow_t', None) is not None:
            hip.check(hip.lib().ecgvit_transpose_bf16_batched(self._wlow.data_ptr(), self._wlow_t.data_ptr(), self._tr_table.data_ptr(),
                                                              self._tr_nmat, self._tr_tiles, hip.stream()), 'transpose_bf16_batched')
        if self._eng is not None and getattr(self._eng, 'fp8', False):
            self._eng.refresh_fp8_weights()

    def set_input_transform(self, transform):
        """f2: give the model RAW records; Normalize / TimeEndPad / TimeOut run fused inside the patch-embed load
        (`transform.FusedInputTransform`). `config.max_signal_length` must be the padded length. None restores the default."""
        self._input_transform = transform
        self._eng = None
        return self

    def set_compute_dtype(self, dtype):
        self.compute_dtype = dtype
        self._eng = None
        return self

    def patch_embed(self, sample_values):
        """Rearrange + Linear of `vit.to_patch_embedding` on its own: (B, C, L) -> (B, n, d) f32."""
        eng = self._engine()
        x = sample_values.contiguous().float()
        B = x.shape[0]
        eng._alloc(B)
        a = eng.act
        hip.check(hip.lib().ecgvit_patch_gather(x.data_ptr(), a['patches'].data_ptr(), B, eng.C, eng.L, eng.P, eng.CP,
                                                hip.code(eng.dtype), hip.stream()), 'patch_gather')
        hip.gemm(hip.GEMM_NT, a['patches'], eng.W['vit.to_patch_embedding.1.weight'], a['tok'], B * eng.n, eng.d, eng.CP,
                 eng.CP, eng.CP, eng.d, epilogue=hip.EPI_BIAS, bias=eng.P32['vit.to_patch_embedding.1.bias'])
        return a['tok'].float().view(B, eng.n, eng.d).clone()

    def attention_probs(self, layer):
        return self._engine().attention_probs(layer).clone()

    def attention_rollout(self, sample_values):
        """The attention map `EcgVitVisualizer.__call__` derives (reference ecg_vit.py:176-194) for ONE (12, L) record: per-layer
        head-averaged attention + identity, row-normalised, multiplied with the layer below, CLS row, scaled by the global maximum.
        Returns (logits (K,), map (layers, n_patch)) as device tensors; plotting stays with the caller."""
        assert sample_values.dim() == 2 and sample_values.size(0) == self.config.num_channels
        was = self.training
        self.eval()
        with torch.no_grad():
            logits = self(sample_values=sample_values.unsqueeze(0).contiguous()).logits[0]
            attn = torch.stack([self._engine().attention_probs(i)[0].mean(dim=0) for i in range(self.config.num_hidden_layers)])
            attn = attn + torch.eye(attn.size(1), device=attn.device)
            attn = attn / attn.sum(dim=-1, keepdim=True)
            res = torch.empty_like(attn)
            res[0] = attn[0]
            for i in range(1, attn.size(0)):
                res[i] = attn[i] @ attn[i - 1]
            res = res[:, 0, 1:]
            res = res / res.max()
        self.train(was)
        return logits, res


def load_trained(model_key='ecg-vit-base', checkpoint_path=None, compute_dtype=torch.float32):
    """`load_trained` of the reference (ecg_vit.py:150-161): build the named config, `torch.load` the `.pt` state_dict the reference's
    trainer wrote (`torch.save(model.state_dict())`, models/train.py:297-300, :319), STRICT load, eval mode. The reference hard-codes
    the path of its own run; here it is an argument."""
    if checkpoint_path is None:
        raise ValueError('checkpoint_path is required (the reference hard-codes a path inside its own model directory)')
    model = EcgVit(config=EcgVitConfig.from_defined(model_key), compute_dtype=compute_dtype)
    ckpt = torch.load(checkpoint_path, map_location='cpu')
    model.load_state_dict(ckpt, strict=True)
    model.eval()
    return model


# ----------------------------------------------------------------------------------------------------------
# masked pre-train objective -- NOT in the reference: the build's own SimMIM-style definition
# ----------------------------------------------------------------------------------------------------------
class _MaskedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wrapper, x, idx, *params):
        enc = wrapper.encoder
        eng = enc._engine()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (enc.training and enc._has_dropout) else 0
        pred, loss = eng.forward_masked(x, idx, training=enc.training, seed=seed)
        enc._fwd_id += 1
        ctx.wrapper, ctx.fwd_id = wrapper, enc._fwd_id
        ctx.set_materialize_grads(False)
        B, m = idx.shape
        return loss.clone().reshape(()), pred.float().view(B, m, -1).clone()

    @staticmethod
    def backward(ctx, gloss, gpred):
        enc = ctx.wrapper.encoder
        if ctx.fwd_id != enc._fwd_id:
            raise RuntimeError('MaskedEcgVit: a later forward overwrote the activations this backward needs')
        if gpred is not None:
            raise NotImplementedError('gradients through the reconstruction output')
        if gloss is None:
            return (None,) * (3 + len(enc._param_list))
        keep, aliased = _grads_living_in_flat_buffer(enc, enc._param_names, enc._param_list)
        enc._engine().backward_masked(gscalar=gloss.contiguous().float().reshape(1))
        return (None, None, None) + _grads_out(enc, enc._param_names, keep, aliased)


class MaskedEcgVit(nn.Module):
    """
    Self-supervised masked-patch pre-training around an `EcgVit` encoder (shares all encoder weights):
      patches (B, n, C*P) --Linear--> tokens ; tokens[b, idx[b, :]] <- mask_token ; += pos_embedding[:, 1:n+1] ;
      transformer trunk on the n tokens (no CLS) ; rows idx --Linear(d, C*P)--> reconstruction ;
      loss = mean |reconstruction - raw masked patches|.
    `mask_idx` (B, m) int32: distinct patch indices per record, generated on the host (`random_mask_indices`) so the
    integer index handling is bit-exact and reproducible.  forward -> ModelOutput(loss, logits=(B, m, C*P) reconstruction).
    Extra state_dict keys (not part of the reference checkpoint): `mask_token`, `to_pixels.{weight,bias}`.
    """

    def __init__(self, encoder: EcgVit, mask_ratio: float = 0.5):
        super().__init__()
        self.encoder = encoder
        c = encoder.config
        d, cp = c.hidden_size, c.num_channels * c.patch_size
        self.mask_ratio = mask_ratio
        self.n_patch = c.max_signal_length // c.patch_size
        self.n_mask = max(1, int(mask_ratio * self.n_patch))
        self.mask_token = nn.Parameter(torch.randn(d))
        self.to_pixels = nn.Linear(d, cp)
        encoder._attach_extra([('pretrain.mask_token', self.mask_token), ('pretrain.to_pixels.weight', self.to_pixels.weight),
                               ('pretrain.to_pixels.bias', self.to_pixels.bias)])

    def random_mask_indices(self, batch, generator=None):
        """(B, m) int32 on the host: per record, the first m entries of a random permutation of the n patches"""
        return torch.stack([torch.randperm(self.n_patch, generator=generator)[:self.n_mask] for _ in range(batch)]).to(torch.int32)

    def check_mask_indices(self, mask_idx, batch):
        """(B, m) integer tensor of DISTINCT patch indices in [0, n_patch) per record, else ValueError: the row gather / scatter
        kernels index with them unchecked (an out-of-range index is an out-of-bounds access, a duplicate a racy scatter)"""
        if mask_idx.dim() != 2 or mask_idx.shape[0] != batch or not 0 < mask_idx.shape[1] <= self.n_patch:
            raise ValueError(f'mask_idx must be (batch={batch}, m) with 0 < m <= {self.n_patch}, got {tuple(mask_idx.shape)}')
        if mask_idx.dtype.is_floating_point or mask_idx.dtype == torch.bool:
            raise ValueError(f'mask_idx must be an integer tensor, got {mask_idx.dtype}')
        if not mask_idx.is_cuda:
            # host indices (what random_mask_indices hands out): checked on the host -- no device round trips, so a training loop that draws its
            # masks per step keeps the host ahead of the device (a device tensor costs three blocking reads below)
            a = np.sort(mask_idx.numpy().astype(np.int64, copy=False), axis=1)
            if a[:, 0].min() < 0 or a[:, -1].max() >= self.n_patch:
                raise ValueError(f'mask_idx entries must lie in [0, {self.n_patch})')
            if (a[:, 1:] == a[:, :-1]).any():
                raise ValueError('mask_idx holds a duplicate patch index inside a record')
            return
        idx = mask_idx.to(torch.int64)
        if int(idx.min()) < 0 or int(idx.max()) >= self.n_patch:
            raise ValueError(f'mask_idx entries must lie in [0, {self.n_patch})')
        srt = torch.sort(idx, dim=1).values
        if bool((srt[:, 1:] == srt[:, :-1]).any()):
            raise ValueError('mask_idx holds a duplicate patch index inside a record')

    def forward(self, sample_values, mask_idx):
        if not sample_values.is_cuda:
            raise RuntimeError('MaskedEcgVit (HIP) runs on an MI355X device only (no CPU fallback)')
        x = sample_values.contiguous().float()
        self.check_mask_indices(mask_idx, x.shape[0])
        idx = mask_idx.to(device=x.device, dtype=torch.int32).contiguous()
        loss, pred = _MaskedFunction.apply(self, x, idx, *self.encoder._param_list)
        return ModelOutput(loss=loss, logits=pred)
