"""
Host-side schedule of the ECG-ViT step over the C-ABI kernels (`include/ecgvit_hip.h`).

This file holds NO arithmetic: it owns the HBM layout (flat f32 parameter / gradient / optimiser buffers,
a bf16 shadow of the weights, per-layer activation slabs kept resident for the backward pass -- 288 GB of
HBM3E makes recomputation pointless at these sizes) and issues the kernels in order on the current HIP
stream.  What each launch replaces in the reference is cited at the call site
(vit-pytorch 0.33.2 `ViT.forward` reached from `ecg_transformer/models/ecg_vit.py:141`).
"""
import math
from collections import OrderedDict

import os

import torch

from . import hip
from .hip import lib, check, ptr, stream, GEMM_NT, GEMM_NN, GEMM_TN
from .hip import EPI_BIAS, EPI_GELU, EPI_GELU_BWD, EPI_RESIDUAL, EPI_DROPOUT, EPI_COLSUM, EPI_GELU_GRAD_AUX, EPI_MUL_AUX

LN_EPS = 1e-5  # nn.LayerNorm default, used by vit_pytorch PreNorm / mlp_head


def _align(n, a=8):
    return (n + a - 1) // a * a


def _numel(shape):
    n = 1
    for v in shape:
        n *= v
    return n


class ParamLayout:
    """name -> (offset, shape) inside one flat f32 buffer; every tensor starts on a 64-B boundary so the same
    offsets address the bf16 shadow on 32-B and the 8-bit shadows on 16-B boundaries (vector loads everywhere)."""

    def __init__(self, named_shapes):
        self.entries = OrderedDict()
        off = 0
        for name, shape in named_shapes:
            n = 1
            for s in shape:
                n *= s
            self.entries[name] = (off, tuple(shape), n)
            off = _align(off + n, 16)
        self.total = off

    def view(self, flat, name):
        off, shape, n = self.entries[name]
        return flat[off:off + n].view(shape)

    def span(self, pred):
        """[lo, hi) element range of the flat buffer covered by the parameters whose name satisfies `pred` (must be contiguous)"""
        sel = [(o, o + c) for n, (o, _, c) in self.entries.items() if pred(n)]
        if not sel:
            return None
        lo, hi = min(a for a, _ in sel), max(b for _, b in sel)
        inside = [n for n, (o, _, c) in self.entries.items() if lo <= o < hi]
        assert all(pred(n) for n in inside), 'bucket is not contiguous in the flat layout'
        return lo, _align(hi, 16)

    def buckets_in_ready_order(self, n_layers):
        """gradient buckets in the order the backward pass completes them: head, layers L-1..0, then embedding (+ extras)"""
        out = [('head', self.span(lambda n: n.startswith('vit.mlp_head.')))]
        for i in reversed(range(n_layers)):
            out.append((f'layer{i}', self.span(lambda n, i=i: n.startswith(f'vit.transformer.layers.{i}.'))))
        out.append(('embed', self.span(lambda n: n in ('vit.pos_embedding', 'vit.cls_token') or n.startswith('vit.to_patch_embedding.'))))
        ex = self.span(lambda n: n.startswith('pretrain.'))
        if ex:
            out.append(('pretrain', ex))
        return [(k, v) for k, v in out if v]


class VitEngine:
    """Forward / backward of EcgVit for one activation dtype (torch.float32 = parity path, torch.bfloat16 =
    throughput path). Caller provides the flat buffers; all activations are allocated here, once per batch size."""

    def __init__(self, *, C, L, P, d, h, f, Ly, K, p_hidden, p_emb, dtype, layout: ParamLayout, fp8_linear=False, saved_ffn_e4m3=None):
        assert L % P == 0, 'Image dimensions must be divisible by the patch size.'  # vit_pytorch's own assertion text
        self.C, self.L, self.P, self.d, self.h, self.f, self.Ly, self.K = C, L, P, d, h, f, Ly, K
        self.n = L // P
        self.N = self.n + 1
        self.dh = d // h
        self.CP = C * P
        self.p_hidden, self.p_emb = float(p_hidden), float(p_emb)
        self.dtype = dtype
        self.layout = layout
        self.scale = self.dh ** -0.5
        if d % 8 or f % 8 or self.CP % 8 or self.dh % 8:
            raise ValueError(f'HIP path needs hidden_size, intermediate_size, head dim and C*P to be multiples of 8 '
                             f'(got d={d}, f={f}, dh={self.dh}, C*P={self.CP})')
        if d > 2048:
            raise ValueError('HIP LayerNorm kernels cover hidden_size <= 2048')
        if h == 1:
            raise NotImplementedError('heads == 1 (vit_pytorch drops to_out) is not covered by the HIP path')
        if dtype == torch.bfloat16:
            if self.dh != 64:
                raise ValueError(f'bf16 fused attention needs head dim 64 (got {self.dh}); use dtype=torch.float32')
            if self.N > 512:
                raise ValueError(f'bf16 fused attention covers <= 512 tokens (got {self.N}); use dtype=torch.float32')
            for nm, pv in (('hidden_dropout_prob', self.p_hidden), ('attention_probs_dropout_prob (the embedding dropout: reference ecg_vit.py:113)', self.p_emb)):
                if 0.0 < pv < 1.0 / 512:
                    # every dropout site of the bf16 path draws 8 random bits per element: p is applied as round(256 p) / 256 (0.1 -> 0.1016)
                    raise ValueError(f'the bf16 path applies dropout in steps of 1/256: {nm}={pv} would round to no dropout; use 0, a value '
                                     f'>= 1/512, or compute_dtype=torch.float32 (exact p)')
        # fp8 Linear operands (BASELINE.json configs[4]): every product of the four block Linears takes 8-bit operands -- forward e4m3 x e4m3,
        # input gradients e5m2 gradients x e4m3 weights, weight gradients e5m2 gradients x e4m3 activations with f32 split-K accumulation
        # (per-tensor scales, delayed for activations and gradients); attention, LayerNorm and the optimiser stay bf16 / f32
        self.fp8 = bool(fp8_linear)
        if self.fp8:
            if dtype != torch.bfloat16:
                raise ValueError('fp8_linear needs the bf16 engine (compute_dtype=torch.bfloat16)')
            if d % 128 or f % 128:
                raise ValueError(f'fp8_linear needs hidden_size and intermediate_size to be multiples of 128 (got d={d}, f={f})')
        # fp8_linear, steady state: four bf16 tensors per layer have 8-bit readers ONLY once their producers emit the 8-bit copies -- the
        # LayerNorm outputs xn1 / xn2 (QKV / FFN-up product + their weight gradients), the FFN-up output hact (FFN-down product + weight
        # gradient), the FFN-down input gradient dh (FFN-up input + weight gradient) and the dropout-masked gradient dxm of the fused LayerNorm
        # backward.  They are then not written at all (ECGVIT_EPI_NO_OUT / NULL y / NULL dxm): ~2.9 GB of stores less per EcgVit-large layer at
        # 256 x 501 tokens.  Needs every reader on its 8-bit kernel: the weight-gradient kernel wants d, f % 256 == 0 and >= 4096 rows.
        # (False: keep writing them -- tests hold the two modes against each other bit for bit.)
        self.fp8_drop_dead_bf16 = self.fp8 and d % 256 == 0 and f % 256 == 0
        # the saved FFN tensor gelu'(pre) x dropout multiplier as e4m3 bytes (ECGVIT_EPI_AUX8; bf16 or 8-bit operands, large A.B^T kernel): private to the
        # FFN-up forward and the FFN-down input gradient, 790 MB per layer at base whose HBM stream costs each launch ~85 us.  False: bf16.
        # `saved_ffn_e4m3` (EcgVit(..., saved_ffn_e4m3=)): None = on for the bf16 engine; False = keep the tensor in bf16
        if saved_ffn_e4m3 and dtype != torch.bfloat16:
            raise ValueError('saved_ffn_e4m3 needs the bf16 engine (the f32 parity path keeps the f32 pre-activation)')
        self.aux8 = dtype == torch.bfloat16 if saved_ffn_e4m3 is None else bool(saved_ffn_e4m3)
        self.B = None
        self._alloc_key = None
        self._pool, self._pool_group, self._pool_B = None, None, 0
        self.T = self.N
        self.act = None
        self.P32 = self.G32 = self.W = None
        self._ws = {}
        self.saved = None
        self.input_transform = None  # FusedInputTransform: forward() then takes RAW (B, C, L_raw) records
        self.on_grads_ready = None   # callback(tag): a gradient bucket ('head' | 'layer{i}' | 'embed' | 'pretrain') is final
        self._tpw = 0                # ecgvit_gemm_desc.tiles_per_workgroup of the launches in flight (set by backward(..., tiles_per_workgroup=))

    def _gemm(self, *a, **kw):
        """every product of this engine: `tiles_per_workgroup` is per-engine, per-pass state (never a process-wide setting)"""
        kw.setdefault('tiles_per_workgroup', self._tpw)
        hip.gemm(*a, **kw)

    # ---------------------------------------------------------------- buffers
    def bind(self, pflat, gflat, wlow=None, wlow_t=None):
        """pflat/gflat: flat f32 params / grads. wlow: flat bf16 shadow of pflat (bf16 engine only); wlow_t: same layout, the trunk's
        Linear weights stored TRANSPOSED (see `transposed_weight_table`), so their input-gradient products run as A . B^T."""
        lay = self.layout
        self.P32 = {k: lay.view(pflat, k) for k in lay.entries}
        self.G32 = {k: lay.view(gflat, k) for k in lay.entries}
        self.WT = {}
        if self.dtype == torch.bfloat16:
            assert wlow is not None and wlow.dtype == torch.bfloat16
            self.W = {k: lay.view(wlow, k) for k in lay.entries}
            if wlow_t is not None:
                for k in self.transposed_weight_names():
                    off, shape = lay.entries[k][0], lay.entries[k][1]
                    self.WT[k] = wlow_t[off:off + shape[0] * shape[1]].view(shape[1], shape[0])
        else:
            self.W = self.P32
        self.device = pflat.device
        if self.fp8:
            assert wlow_t is not None, 'fp8_linear runs the input gradients against the transposed weight shadows'
            names = self.transposed_weight_names()
            assert len(names) == 4 * self.Ly, 'fp8_linear: every block Linear must be large enough for the 256^2 kernel'
            self._wlow, self._wlow_t = wlow, wlow_t
            self.w8 = torch.zeros(lay.total, dtype=torch.uint8, device=self.device)
            self.w8t = torch.zeros(lay.total, dtype=torch.uint8, device=self.device)
            self.w8_index = {k: i for i, k in enumerate(names)}
            self.w8_table = torch.tensor([[lay.entries[k][0], lay.entries[k][2]] for k in names], dtype=torch.int64, device=self.device)
            self.w8_count = max(lay.entries[k][2] for k in names)
            self.w8_scale = torch.zeros(len(names), dtype=torch.float32, device=self.device)
            self.w8_amax = torch.zeros(len(names), dtype=torch.float32, device=self.device)
            self.W8, self.WT8 = {}, {}
            for k in names:
                off, (r, c), n = lay.entries[k]
                self.W8[k] = self.w8[off:off + n].view(r, c)
                self.WT8[k] = self.w8t[off:off + n].view(c, r)
            # activation / gradient sites, 8 per layer: e4m3 xn1, attn, xn2, hact ; e5m2 dY(ffn-down), dh, dY(out), dqkv
            ns = 8 * self.Ly
            self.f8_scale = torch.zeros(ns, dtype=torch.float32, device=self.device)
            self.f8_amax = torch.zeros(ns, dtype=torch.float32, device=self.device)
            self.f8_fmt = torch.tensor(([hip.FP8_E4M3] * 4 + [hip.BF8_E5M2] * 4) * self.Ly, dtype=torch.int32, device=self.device)
            self._f8_seen = set()
            self._f8_last_training, self._f8_scale_train = None, None

    # ---------------------------------------------------------------- fp8 operand path
    def refresh_fp8_weights(self):
        """e4m3 shadows of the block Linears' weights and of their transposes, one scale per matrix from its current amax: three
        launches over the flat bf16 shadows (call after the optimiser rewrote them)"""
        l, st = lib(), stream()
        nm = len(self.w8_index)
        check(l.ecgvit_fp8_amax(ptr(self._wlow), ptr(self.w8_table), nm, self.w8_count, ptr(self.w8_amax), st), 'fp8_amax')
        check(l.ecgvit_fp8_scale_update(ptr(self.w8_scale), ptr(self.w8_amax), nm, None, hip.FP8_E4M3, st), 'fp8_scale_update')
        for src, dst in ((self._wlow, self.w8), (self._wlow_t, self.w8t)):
            check(l.ecgvit_fp8_quantize(ptr(src), ptr(dst), ptr(self.w8_table), nm, self.w8_count, hip.FP8_E4M3, ptr(self.w8_scale), None, st),
                  'fp8_quantize')

    def fp8_begin_step(self, training=True):
        """delayed scaling: the scales of this pass come from the amax the previous pass's quantise kernels accumulated.
        One exception: a TRAINING pass that follows EVAL passes resumes from the scales the last training pass left (snapshotted when the first eval
        pass began; the eval passes' amax is discarded) -- eval activations carry no dropout, training tensors are rescaled by 1 / (1 - p) (the FFN
        hidden activation most of all), and scales are amax / format-max with no headroom: scales taken from an eval pass would saturate the
        first training step after every evaluation."""
        if self._f8_seen:
            if training and self._f8_last_training is False and self._f8_scale_train is not None:
                self.f8_scale.copy_(self._f8_scale_train)
                self.f8_amax.zero_()
            else:
                check(lib().ecgvit_fp8_scale_update(ptr(self.f8_scale), ptr(self.f8_amax), self.f8_scale.numel(), ptr(self.f8_fmt), 0, stream()),
                      'fp8_scale_update')
                if not training and self._f8_last_training:
                    self._f8_scale_train = self.f8_scale.clone()   # train -> eval: what the last training pass's amax gave (8 floats per layer)
        self._f8_last_training = bool(training)

    def _quant(self, site, x, count, out=None):
        """x (bf16, `count` elements) -> `out` (a layer's persistent e4m3 copy: the weight-gradient product reads it again in the
        backward pass) or the shared 8-bit scratch, in the site's format; returns (8-bit view, scale pointer tensor)"""
        l, st = lib(), stream()
        sc, am = self.f8_scale[site:site + 1], self.f8_amax[site:site + 1]
        fmt = hip.FP8_E4M3 if site % 8 < 4 else hip.BF8_E5M2
        if site not in self._f8_seen:   # first use: no history yet -> scale from this tensor's own amax (one extra read)
            check(l.ecgvit_fp8_amax(ptr(x), None, 1, count, ptr(am), st), 'fp8_amax')
            check(l.ecgvit_fp8_scale_update(ptr(sc), ptr(am), 1, None, fmt, st), 'fp8_scale_update')
            self._f8_seen.add(site)
        q = out if out is not None else self.act['q8'][:count]
        check(l.ecgvit_fp8_quantize(ptr(x), ptr(q), None, 1, count, fmt, ptr(sc), ptr(am), st), 'fp8_quantize')
        return q, sc

    def _emit8(self, kw, site, ld, out=None, only8=False):
        """ask an 8-bit product's epilogue to also write the 8-bit copy of its output that the next product (site `site`) consumes --
        possible once that site has a scale (from the second pass on); `out`: where (default: the q8b scratch); only8: the bf16 output has
        no reader then (EPI_NO_OUT); returns True when armed"""
        if site not in self._f8_seen:
            return False
        kw['epilogue'] = kw.get('epilogue', 0) | hip.EPI_QUANT_OUT | (hip.EPI_NO_OUT if only8 else 0)
        kw.update(q8_out=out if out is not None else self.act['q8b'], ldq8=ld, q8_scale=self.f8_scale[site:site + 1], q8_amax=self.f8_amax[site:site + 1],
                  q8_format=hip.FP8_E4M3 if site % 8 < 4 else hip.BF8_E5M2)
        return True

    def _aux8(self, M):
        """the FFN-wide launches over M token rows take the large A.B^T kernel (the mirror of ecgvit_gemm_nt_applicable for [M, f] x K = d), so the
        saved tensor may be e4m3 bytes; the library rejects the flag loudly if this ever disagrees with its own dispatch"""
        return (self.aux8 and bool(self.WT) and M >= 2048 and self.f >= 128 and self.f % 8 == 0 and self.d % 64 == 0 and self.d >= 192
                and (M + 256) * self.f * 2 < 2 ** 31)   # (self.WT: the input gradient runs as A.B^T against the transposed shadow)

    def _only8(self, M):
        """the bf16 copies with 8-bit readers only may be left unwritten in a pass over M token rows (see `fp8_drop_dead_bf16`)"""
        return self.fp8 and self.fp8_drop_dead_bf16 and M >= 4096

    def _bf16_reader(self, M, what):
        """called by every bf16 fallback of a block product: when `_only8(M)` holds, the producers have left xn1 / xn2 / hact / dh / dxm
        UNWRITTEN on the promise that every reader takes its 8-bit kernel -- a reader that falls back to bf16 would consume stale
        buffers from an earlier step and return wrong gradients without any error.  The promise is written out in several places
        (`_only8`, `_wgrad`, `_dgrad`, the library's applicability checks); if they ever drift apart, fail here"""
        if self._only8(M):
            raise RuntimeError(f'fp8_linear: {what} fell back to its bf16 kernel over {M} rows while the bf16 operand copies are not '
                               f'written (fp8_drop_dead_bf16); the 8-bit applicability conditions have drifted apart')

    def _linear(self, site, A, name, C, M, N, K, a8=None, emit_site=None, emit_to=None, prequant=False, emit_only8=False, **kw):
        """C = epilogue(A . W^T) for block Linear `name`: bf16 operands, or (fp8_linear) A quantised to e4m3 against the e4m3 shadow.
        a8: this layer's persistent e4m3 copy of A (written here, or already by A's producer when `prequant`; the weight-gradient
        product of the backward pass reads it again); emit_site / emit_to: the site that consumes C next and its persistent copy
        (then written by this epilogue).  Returns True when the 8-bit copy of C was emitted."""
        if not self.fp8 or M < 2048:    # the 8-bit kernel covers the large products only: small batches run bf16
            self._bf16_reader(M, 'Linear ' + name)
            self._gemm(GEMM_NT, A, self.W[name], C, M, N, K, K, K, N, **kw)
            return False
        if prequant:   # the producer (LayerNorm forward, a GEMM epilogue) already wrote A's 8-bit copy into a8
            q, sc = a8, self.f8_scale[site:site + 1]
        else:
            q, sc = self._quant(site, A, M * K, out=a8)
        emitted = emit_site is not None and self._emit8(kw, emit_site, N, out=emit_to, only8=emit_only8)
        mi = self.w8_index[name]
        self._gemm(GEMM_NT, q, self.W8[name], None if kw.get('epilogue', 0) & hip.EPI_NO_OUT else C, M, N, K, K, K, N, fp8_format=hip.FP8_E4M3,
                   scale_a=sc, scale_b=self.w8_scale[mi:mi + 1], **kw)
        return emitted

    def transposed_weight_names(self):
        """Linear weights of the transformer blocks whose dgrad is large enough for the 256^2 forward kernel (K % 64 == 0, N >= 256)"""
        out = []
        for i in range(self.Ly):
            lp = f'vit.transformer.layers.{i}.'
            for k in ('0.fn.to_qkv.weight', '0.fn.to_out.0.weight', '1.fn.net.0.weight', '1.fn.net.3.weight'):
                rows, cols = self.layout.entries[lp + k][1]
                if rows % 64 == 0 and rows >= 192 and cols >= 256 and cols % 8 == 0:
                    out.append(lp + k)
        return out

    def transposed_weight_table(self, device):
        """(table tensor int64 [nmat, 4] on `device`, nmat, ntiles) for ecgvit_transpose_bf16_batched"""
        rows_, t = [], 0
        for k in self.transposed_weight_names():
            off, (r, c) = self.layout.entries[k][0], self.layout.entries[k][1]
            rows_.append([off, r, c, t])
            t += ((r + 63) // 64) * ((c + 63) // 64)
        if not rows_:
            return None, 0, 0
        return torch.tensor(rows_, dtype=torch.int64, device=device), len(rows_), t

    def _grad8(self, site, dY, count, prequant=False):
        """(e5m2 copy of the gradient dY entering site `site`, its scale) for the site's two backward products -- the input gradient
        dY . W and the weight gradient dY^T . X; prequant: its producer already wrote the copy ('q8': LayerNorm backward into the
        operand scratch, True: a GEMM epilogue into q8b).  None when the 8-bit path does not apply."""
        if prequant:
            return self.act['q8' if prequant == 'q8' else 'q8b'][:count], self.f8_scale[site:site + 1]
        # (not prequant: the producer did not emit the copy and therefore DID write its bf16 output -- `_emit8` / the q8 LayerNorm branch arm
        # "no bf16 output" and "8-bit copy emitted" together, and report it through the return value that became `prequant`)
        return self._quant(site, dY, count)

    def _dgrad(self, dY, name, dX, M, kin, nout, site=None, emit_site=None, pre=None, emit_only8=False, **kw):
        """dX[M, kin] = dY[M, nout] . W[nout, kin]: on the forward kernel against the transposed shadow when there is one.
        pre: (8-bit copy of dY, scale) from `_grad8` (fp8_linear).  Returns True when the 8-bit copy of dX was emitted for `emit_site`
        (emit_only8: and dX itself left unwritten)."""
        if self.fp8 and pre is not None and name in self.w8_index:
            q, sc = pre
            emitted = emit_site is not None and self._emit8(kw, emit_site, kin, only8=emit_only8)
            mi = self.w8_index[name]
            self._gemm(GEMM_NT, q, self.WT8[name], None if kw.get('epilogue', 0) & hip.EPI_NO_OUT else dX, M, kin, nout, nout, nout, kin,
                       fp8_format=hip.BF8_E5M2, scale_a=sc, scale_b=self.w8_scale[mi:mi + 1], **kw)
            return emitted
        if self.fp8:
            self._bf16_reader(M, 'input gradient of ' + name)
        wt = self.WT.get(name)
        if wt is not None and M >= 2048:
            self._gemm(GEMM_NT, dY, wt, dX, M, kin, nout, nout, nout, kin, **kw)
        else:
            self._gemm(GEMM_NN, dY, self.W[name], dX, M, kin, nout, nout, kin, kin, **kw)

    def _alloc(self, B, masked=False, m=0):
        """Activation slabs for a pass over B records.  Every slab's leading dimension is proportional to B, so ONE pool serves every batch size
        through prefix views: a loop that alternates train (B = 512) and eval (B = 64) batches, or ends an epoch on a short batch, re-slices instead
        of freeing and re-requesting ~40 GB (base) from the allocator on each switch.  The pool grows PER SLAB and never shrinks inside a token
        geometry; it is dropped only when the objective changes (supervised <-> masked, or another mask count)."""
        key = (B, masked, m, self._aux8(B * (self.n if masked else self.N)))
        if self._alloc_key == key and self.act is not None:
            return
        self._alloc_key = key
        self.T = self.n if masked else self.N   # tokens per record: no CLS row in the masked-pretrain trunk
        spec = self._act_spec(B, masked, m)
        group = (masked, m)
        if self._pool_group != group:
            self.act = self._pool = None           # another token geometry: nothing of the old pool fits
            self._pool_group, self._pool_B = group, 0
        pool = self._pool if self._pool is not None else {}
        self._pool = pool
        self._pool_B = max(B, self._pool_B)
        # Grow-only and PER SLAB inside a group: a slab is re-requested only when this pass needs more bytes of it than the pool holds (a
        # workspace size need not be monotone in B), and slabs the current spec does not name (the e4m3 operand copies of an fp8 model while a
        # short batch runs on the bf16 kernels) stay in the pool for the pass that wants them again.  A slab whose ELEMENT TYPE depends on the
        # pass -- `hpre`: e4m3 bytes over >= 2048 token rows, the activation type below that -- is pooled as bytes and viewed per pass, so a
        # batch that crosses the boundary (a short epoch remainder) re-slices like any other
        for k, (sh, dt) in spec.items():
            pdt = torch.uint8 if k.endswith('.hpre') else dt
            need = _numel(sh) * (dt.itemsize if pdt != dt else 1)
            have = pool.get(k)
            if have is None or have.dtype != pdt or have.numel() < need:
                if self.act is not None:
                    self.act = None                # views of the slab being replaced die with it
                pool[k] = None                     # (drop the old slab before asking for the new one)
                pool[k] = torch.empty(need, device=self.device, dtype=pdt)
        a, layers = {}, [dict() for _ in range(self.Ly)]
        for k, (sh, dt) in spec.items():
            if pool[k].dtype != dt:                # byte slab viewed in this pass's element type
                v = pool[k][:_numel(sh) * dt.itemsize].view(dt).view(sh)
            else:
                v = pool[k][:_numel(sh)].view(sh)
            if k[0] == 'L' and '.' in k:
                i, kk = k[1:].split('.', 1)
                layers[int(i)][kk] = v
            else:
                a[k] = v
        a['layers'] = layers
        self.act, self.B = a, B

    def _act_spec(self, B, masked, m):
        """name -> (shape, dtype) of every activation / scratch slab of a pass over B records (layer slabs as 'L{i}.{name}')"""
        T = self.dtype
        f32, u8 = torch.float32, torch.uint8
        N = self.n if masked else self.N
        M, Mp = B * N, B * self.n
        d, f, h = self.d, self.f, self.h
        sp = OrderedDict()
        sp.update(patches=((Mp, self.CP), T), tok=((Mp, d), T), x0=((M, d), T))
        for i in range(self.Ly):
            l = dict(mean1=((M,), f32), rstd1=((M,), f32), xn1=((M, d), T), qkv=((M, 3 * d), T), attn=((M, d), T), x1=((M, d), T),
                     mean2=((M,), f32), rstd2=((M,), f32), xn2=((M, d), T), hpre=((M, f), u8 if self._aux8(M) else T), hact=((M, f), T), x2=((M, d), T))
            if T == torch.float32:
                l['probs'] = ((B * h * N * N,), T)
            else:
                l['lse'] = ((B * h * N,), f32)
            if self.fp8 and M >= 2048:
                # e4m3 copies of the four Linear inputs, kept for the backward pass: the 8-bit weight-gradient products read them again
                # (one byte per element next to the two of the bf16 tensors: +0.9 GB per layer for large at 256 x 501 tokens)
                l.update(xn1_8=((M * d,), u8), attn_8=((M * d,), u8), xn2_8=((M * d,), u8), hact_8=((M * f,), u8))
            for k, v in l.items():
                sp[f'L{i}.{k}'] = v
        sp.update(logits=((B, self.K), f32), xhat=((B, d), f32), hrstd=((B,), f32), loss_elem=((B, self.K), f32), loss_mean=((1,), f32),
                  dlogits=((B, self.K), f32))
        # backward scratch (shared by all layers)
        sp.update(dxa=((M, d), T), dxb=((M, d), T), dxn=((M, d), T), dqkv=((M, 3 * d), T), dattn=((M, d), T), dh=((M, f), T),
                  dtok=((Mp, d), T), dxm=((M, d), T))
        if T == torch.float32:
            sp.update(pd=((B * h * N * N,), T), dp=((B * h * N * N,), T))
        l = lib()
        ws = max(l.ecgvit_layernorm_bwd_workspace(M, d), l.ecgvit_colsum_workspace(M, max(f, 3 * d)), 8 * ((M + 255) // 256) * f, 4096)
        if T == torch.bfloat16:
            for (mm, nn) in ((d, f), (f, d), (d, d), (3 * d, d), (d, self.CP)):
                ws = max(ws, hip.gemm_workspace_bytes(GEMM_TN, T, mm, nn, M))
        if masked:
            sp.update(flag=((Mp,), u8), rows=((B * m, d), T), pred=((B * m, self.CP), T), target=((B * m, self.CP), T),
                      dpred=((B * m, self.CP), T), drows=((B * m, d), T), dmasked=((Mp, d), T), mloss=((1,), f32), l1part=((1024,), f32))
        sp['ws'] = ((ws,), u8)
        if self.fp8:
            sp['q8'] = ((M * max(f, 3 * d),), u8)   # one quantised operand at a time
            sp['q8b'] = ((M * f,), u8)              # 8-bit copies written by a producing epilogue
        return sp

    # ---------------------------------------------------------------- small launch helpers
    def _ln_fwd(self, x, g, b, y, mean, rstd, rows, q8_site=None, y8=None):
        """LayerNorm forward; q8_site (fp8_linear): also write the e4m3 copy of y into the operand scratch for the Linear that consumes
        it (returns True), once that site has a scale and when the exact-fit kernel covers d"""
        d = self.d
        if (self.fp8 and q8_site is not None and q8_site in self._f8_seen and rows >= 2048 and d % 256 == 0
                and d // 64 in (4, 8, 12, 16, 24, 32)):
            # (y has 8-bit readers only when it has a persistent 8-bit copy for the weight gradient: see fp8_drop_dead_bf16)
            check(lib().ecgvit_layernorm_fwd_q8(ptr(x), ptr(g), ptr(b), None if (y8 is not None and self._only8(rows)) else ptr(y), ptr(mean), ptr(rstd), rows, d, LN_EPS,
                                                ptr(y8 if y8 is not None else self.act['q8']),
                                                ptr(self.f8_scale[q8_site:q8_site + 1]), ptr(self.f8_amax[q8_site:q8_site + 1]), stream()),
                  'layernorm_fwd_q8')
            return True
        check(lib().ecgvit_layernorm_fwd(ptr(x), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, d, LN_EPS,
                                         hip.code(self.dtype), stream()), 'layernorm_fwd')
        return False

    def _ln_bwd(self, dy, x, g, mean, rstd, dres, dx, dg, db, rows):
        check(lib().ecgvit_layernorm_bwd(ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db),
                                         ptr(self.act['ws']), rows, self.d, hip.code(self.dtype), stream()), 'layernorm_bwd')

    def _ln_bwd_fused(self, dy, x, g, mean, rstd, dres, dx, dg, db, rows, dxm, dcolsum, p, seed, q8_site=None):
        """LayerNorm backward that also emits, for the NEXT stage, the dropout-masked copy of dx and its column sums; q8_site
        (fp8_linear): also the e5m2 copy of that gradient into the operand scratch for the input-gradient product that consumes it
        (returns True), once the site has a scale and when the exact-fit kernel covers d"""
        d = self.d
        if (self.fp8 and q8_site is not None and q8_site in self._f8_seen and rows >= 2048 and d // 64 in (4, 8, 12, 16, 24, 32) and d % 64 == 0
                and rows * d < 2 ** 31):
            check(lib().ecgvit_layernorm_bwd_fused_q8(ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db),
                                                      ptr(self.act['ws']), rows, d, None if self._only8(rows) else ptr(dxm), ptr(dcolsum), p, seed, ptr(self.act['q8']),
                                                      ptr(self.f8_scale[q8_site:q8_site + 1]), ptr(self.f8_amax[q8_site:q8_site + 1]), stream()),
                  'layernorm_bwd_fused_q8')
            return True
        check(lib().ecgvit_layernorm_bwd_fused(ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db),
                                               ptr(self.act['ws']), rows, self.d, ptr(dxm), ptr(dcolsum), p, seed,
                                               hip.code(self.dtype), stream()), 'layernorm_bwd_fused')
        return False

    def _colsum(self, x, ld, out, M, N):
        check(lib().ecgvit_colsum(ptr(x), ld, ptr(out), ptr(self.act['ws']), M, N, hip.code(self.dtype), stream()), 'colsum')

    def _drop_apply(self, src, dst, count, p, seed):
        check(lib().ecgvit_dropout_apply(ptr(src), ptr(dst), count, p, seed, hip.code(self.dtype), stream()), 'dropout_apply')

    def _wgrad(self, dY, X, name, Mout, Nin, rows, pre=None, x8=None, xsite=None):
        """dW[Mout, Nin] = dY[rows, Mout]^T . X[rows, Nin]  -> f32 gradient view (overwritten).  fp8_linear: pre = (e5m2 copy of dY,
        scale) from `_grad8`, x8 / xsite = the layer's persistent e4m3 copy of X and its site (scale): the product then runs on the
        8-bit streaming kernel (both operands k-major, transposed 8-bit LDS reads)"""
        if pre is not None and x8 is not None and Mout % 256 == 0 and Nin % 256 == 0 and rows >= 4096:
            q, sc = pre
            self._gemm(GEMM_TN, q, x8, self.G32[name], Mout, Nin, rows, Mout, Nin, Nin, workspace=self.act['ws'], fp8_format=hip.BF8_E5M2,
                       scale_a=sc, scale_b=self.f8_scale[xsite:xsite + 1])
            return
        if self.fp8 and name.startswith('vit.transformer.'):
            self._bf16_reader(rows, 'weight gradient of ' + name)
        self._gemm(GEMM_TN, dY, X, self.G32[name], Mout, Nin, rows, Mout, Nin, Nin, workspace=self.act['ws'])

    # ---------------------------------------------------------------- forward
    def _patch_embed(self, x, B):
        a, W, T = self.act, self.W, hip.code(self.dtype)
        pre = 'vit.'
        # a4: patch Rearrange (integer gather) + Linear(C*P, d)   [+ f2: Normalize / TimeEndPad / TimeOut fused into the load]
        xf = self.input_transform
        if xf is not None:
            mean, inv_std = xf.device_stats(x.device)
            t0 = tl = None
            if xf.timeout and self.saved is not None and self.saved.get('training', True):
                t0, tl = xf.draw_timeout(B, self.L, x.device)
            self._xf_keep = (mean, inv_std, t0, tl)   # keep the int32 spans alive until the kernel has run
            check(lib().ecgvit_patch_gather_transform(ptr(x), ptr(a['patches']), B, self.C, x.shape[2], self.L, self.P, self.CP, ptr(mean),
                                                      ptr(inv_std), ptr(t0), ptr(tl), T, stream()), 'patch_gather_transform')
        else:
            check(lib().ecgvit_patch_gather(ptr(x), ptr(a['patches']), B, self.C, self.L, self.P, self.CP, T, stream()), 'patch_gather')
        self._gemm(GEMM_NT, a['patches'], W[pre + 'to_patch_embedding.1.weight'], a['tok'], B * self.n, self.d, self.CP, self.CP,
                 self.CP, self.d, epilogue=EPI_BIAS, bias=self.P32[pre + 'to_patch_embedding.1.bias'])

    def _trunk_fwd(self, B, ph, seed):
        """L x { x = Attn(LN(x)) + x ; x = FF(LN(x)) + x } on act['x0'] ([B*T, d]); returns the output slab"""
        a, W, T = self.act, self.W, hip.code(self.dtype)
        l, st = lib(), stream()
        d, f, h, dh, N = self.d, self.f, self.h, self.dh, self.T
        M = B * N
        pre = 'vit.'
        X = a['x0']
        for i, L in enumerate(a['layers']):
            lp = f'{pre}transformer.layers.{i}.'
            s0 = seed + 100 * (i + 1)
            # a6/a7: PreNorm(Attention)
            f8 = self.fp8 and M >= 2048
            q1 = self._ln_fwd(X, self.P32[lp + '0.norm.weight'], self.P32[lp + '0.norm.bias'], L['xn1'], L['mean1'], L['rstd1'], M, q8_site=8 * i,
                              y8=L.get('xn1_8'))
            self._linear(8 * i + 0, L['xn1'], lp + '0.fn.to_qkv.weight', L['qkv'], M, 3 * d, d, a8=L.get('xn1_8'), prequant=q1)
            qa = False   # fp8_linear: the attention kernel wrote the e4m3 copy of its output itself
            if self.dtype == torch.bfloat16:
                if f8 and (8 * i + 1) in self._f8_seen:
                    check(l.ecgvit_attention_fwd_q8(ptr(L['qkv']), ptr(L['attn']), ptr(L['lse']), B, N, h, dh, self.scale, ph, s0 + 1, ptr(L['attn_8']),
                                                    ptr(self.f8_scale[8 * i + 1:8 * i + 2]), ptr(self.f8_amax[8 * i + 1:8 * i + 2]), st), 'attention_fwd_q8')
                    qa = True
                else:
                    check(l.ecgvit_attention_fwd(ptr(L['qkv']), ptr(L['attn']), ptr(L['lse']), B, N, h, dh, self.scale, ph, s0 + 1,
                                                 T, st), 'attention_fwd')
            else:
                self._attn_fwd_f32(L, B, ph, s0 + 1)
            epi = EPI_BIAS | EPI_RESIDUAL | (EPI_DROPOUT if ph > 0 else 0)
            self._linear(8 * i + 1, L['attn'], lp + '0.fn.to_out.0.weight', L['x1'], M, d, d, a8=L.get('attn_8'), prequant=qa, epilogue=epi,
                         bias=self.P32[lp + '0.fn.to_out.0.bias'], residual=X, ldr=d, dropout_p=ph, seed=s0 + 2)
            # a6/a8: PreNorm(FeedForward): Linear -> GELU(erf) -> Dropout -> Linear -> Dropout, + residual
            q2 = self._ln_fwd(L['x1'], self.P32[lp + '1.norm.weight'], self.P32[lp + '1.norm.bias'], L['xn2'], L['mean2'], L['rstd2'], M,
                              q8_site=8 * i + 2, y8=L.get('xn2_8'))
            # bf16 path: the saved tensor is gelu'(pre) * dropout multiplier (not the pre-activation): the backward of this site is then
            # one multiply in the input-gradient GEMM's epilogue -- no erf, no mask hash; the f32 parity path keeps the pre-activation
            epi = EPI_BIAS | EPI_GELU | (EPI_DROPOUT if ph > 0 else 0) | (EPI_GELU_GRAD_AUX if self.dtype == torch.bfloat16 else 0)
            if self._aux8(M):
                epi |= hip.EPI_AUX8
            hq = self._linear(8 * i + 2, L['xn2'], lp + '1.fn.net.0.weight', L['hact'], M, f, d, a8=L.get('xn2_8'), emit_site=8 * i + 3,
                              emit_to=L.get('hact_8'), prequant=q2, emit_only8=self._only8(M), epilogue=epi,
                              bias=self.P32[lp + '1.fn.net.0.bias'], aux=L['hpre'], ldaux=f, dropout_p=ph, seed=s0 + 3)
            epi = EPI_BIAS | EPI_RESIDUAL | (EPI_DROPOUT if ph > 0 else 0)
            self._linear(8 * i + 3, L['hact'], lp + '1.fn.net.3.weight', L['x2'], M, d, f, a8=L.get('hact_8'), prequant=hq, epilogue=epi,
                         bias=self.P32[lp + '1.fn.net.3.bias'], residual=L['x1'], ldr=d, dropout_p=ph, seed=s0 + 4)
            X = L['x2']
        return X

    def forward(self, x, labels=None, weight=None, training=True, seed=0, want_mean=True):
        """x: (B, C, L) f32 contiguous device tensor. Returns (logits (B,K) f32, loss_elem (B,K) f32 | None, loss_mean (1,) | None)."""
        B = x.shape[0]
        assert x.shape[1] == self.C and x.dtype == torch.float32 and x.is_contiguous()
        if self.input_transform is None:
            assert x.shape[2] == self.L
        else:
            assert self.input_transform.padded_length(x.shape[2]) == self.L, 'config.max_signal_length must be the padded length'
        self._alloc(B)
        a, T = self.act, hip.code(self.dtype)
        l, st = lib(), stream()
        d, N, n = self.d, self.N, self.n
        ph = self.p_hidden if training else 0.0
        pe = self.p_emb if training else 0.0
        self.saved = dict(B=B, ph=ph, pe=pe, seed=seed, labels=labels, weight=weight, masked=False, training=training)
        if self.fp8:
            # EVERY forward, eval included, starts from the scales of the pass before it (delayed scaling with a history of one pass): an
            # inference-only model otherwise keeps its first batch's scales forever and clamps larger activations silently.  No backward can be
            # waiting for the old scales: a later forward overwrites the activations that backward reads (one live graph per model -- the
            # autograd node raises on a stale backward)
            self.fp8_begin_step(training)
        pre = 'vit.'
        self._patch_embed(x, B)
        # a5: cat CLS, += pos_embedding[:, :n+1], emb dropout
        check(l.ecgvit_embed_finish(ptr(a['tok']), ptr(self.P32[pre + 'cls_token']), ptr(self.P32[pre + 'pos_embedding']),
                                    ptr(a['x0']), B, n, d, pe, seed + 1, T, st), 'embed_finish')
        X = self._trunk_fwd(B, ph, seed)
        self.saved['xL'] = X
        # a10: x[:, 0] -> LayerNorm -> Linear(d, K)
        check(l.ecgvit_head_fwd(ptr(X), N, ptr(self.P32[pre + 'mlp_head.0.weight']), ptr(self.P32[pre + 'mlp_head.0.bias']),
                                ptr(self.P32[pre + 'mlp_head.1.weight']), ptr(self.P32[pre + 'mlp_head.1.bias']),
                                ptr(a['logits']), ptr(a['xhat']), ptr(a['hrstd']), B, d, self.K, LN_EPS, T, st), 'head_fwd')
        if labels is None:
            return a['logits'], None, None
        # a11: BCEWithLogitsLoss
        check(l.ecgvit_bce_fwd(ptr(a['logits']), ptr(labels), ptr(weight), ptr(a['loss_elem']),
                               ptr(a['loss_mean']) if want_mean else None, B * self.K, st), 'bce_fwd')
        return a['logits'], a['loss_elem'], (a['loss_mean'] if want_mean else None)

    # ---------------------------------------------------------------- masked pre-train objective
    def forward_masked(self, x, idx, training=True, seed=0):
        """SimMIM-style step (build's own definition; absent from the reference): tokens = Linear(patches); masked tokens <-
        mask_token; + pos[1:n+1]; trunk on n tokens (no CLS); masked rows -> Linear(d, C*P); L1 vs the raw masked patches.
        x (B,C,L) f32; idx (B,m) int32 distinct patch indices per record. Returns (pred (B*m, C*P), loss (1,) f32)."""
        B, m = idx.shape
        assert idx.dtype == torch.int32 and idx.is_contiguous() and 0 < m <= self.n
        self._alloc(B, masked=True, m=m)
        a, W, T = self.act, self.W, hip.code(self.dtype)
        l, st = lib(), stream()
        d, n = self.d, self.n
        ph = self.p_hidden if training else 0.0
        pe = self.p_emb if training else 0.0
        self.saved = dict(B=B, ph=ph, pe=pe, seed=seed, masked=True, idx=idx, m=m, training=training)
        if self.fp8:
            self.fp8_begin_step(training)
        self._patch_embed(x, B)
        check(l.ecgvit_mask_embed_finish(ptr(a['tok']), ptr(self.P32['pretrain.mask_token']), ptr(self.P32['vit.pos_embedding']),
                                         ptr(idx), ptr(a['x0']), ptr(a['flag']), B, n, m, d, T, st), 'mask_embed_finish')
        if pe > 0:
            self._drop_apply(a['x0'], a['x0'], B * n * d, pe, seed + 1)
        X = self._trunk_fwd(B, ph, seed)
        self.saved['xL'] = X
        check(l.ecgvit_gather_rows(ptr(X), ptr(idx), ptr(a['rows']), B, n, m, d, d, d, T, st), 'gather_rows')
        self._gemm(GEMM_NT, a['rows'], W['pretrain.to_pixels.weight'], a['pred'], B * m, self.CP, d, d, d, self.CP, epilogue=EPI_BIAS,
                 bias=self.P32['pretrain.to_pixels.bias'])
        check(l.ecgvit_gather_rows(ptr(a['patches']), ptr(idx), ptr(a['target']), B, n, m, self.CP, self.CP, self.CP, T, st), 'gather_rows')
        # L1 loss and d(loss)/d(pred) in one pass (upstream gradient 1; backward_masked re-runs it for any other upstream)
        check(l.ecgvit_l1_loss_fwd_bwd(ptr(a['pred']), ptr(a['target']), ptr(a['mloss']), ptr(a['dpred']), None, ptr(a['l1part']), B * m, self.CP,
                                       self.CP, T, st), 'l1_loss')
        return a['pred'], a['mloss']

    def backward_masked(self, gscalar=None, tiles_per_workgroup=0):
        """loss + every gradient of the masked objective (the L1 kernel produces loss and dpred in one pass).
        tiles_per_workgroup: as `backward`"""
        self._tpw = int(tiles_per_workgroup)
        try:
            self._backward_masked(gscalar)
        finally:
            self._tpw = 0

    def _backward_masked(self, gscalar):
        a, W, T = self.act, self.W, hip.code(self.dtype)
        l, st = lib(), stream()
        sv = self.saved
        B, m, idx, pe, seed = sv['B'], sv['m'], sv['idx'], sv['pe'], sv['seed']
        d, n = self.d, self.n
        G = self.G32
        if gscalar is not None:
            check(l.ecgvit_l1_loss_fwd_bwd(ptr(a['pred']), ptr(a['target']), ptr(a['mloss']), ptr(a['dpred']), ptr(gscalar), ptr(a['l1part']), B * m,
                                           self.CP, self.CP, T, st), 'l1_loss')
        # the classification head does not take part: its gradients are zero for this objective -- known at once, so its bucket is
        # released FIRST and its exchange overlaps the whole backward pass (buckets complete in the order head, layers L-1..0, embed,
        # pretrain, as in the supervised pass)
        for k in ('vit.mlp_head.0.weight', 'vit.mlp_head.0.bias', 'vit.mlp_head.1.weight', 'vit.mlp_head.1.bias', 'vit.cls_token'):
            G[k].zero_()
        self._ready('head')
        self._colsum(a['dpred'], self.CP, G['pretrain.to_pixels.bias'], B * m, self.CP)
        self._wgrad(a['dpred'], a['rows'], 'pretrain.to_pixels.weight', self.CP, d, B * m)
        self._gemm(GEMM_NN, a['dpred'], W['pretrain.to_pixels.weight'], a['drows'], B * m, d, self.CP, self.CP, d, d)
        dX = a['dxa']
        dX.zero_()
        check(l.ecgvit_scatter_rows(ptr(a['drows']), ptr(idx), ptr(dX), B, n, m, d, d, d, T, st), 'scatter_rows')
        dX = self._trunk_bwd(dX, a['dxb'])
        if pe > 0:
            self._drop_apply(dX, dX, B * n * d, pe, seed + 1)
        check(l.ecgvit_mask_embed_bwd(ptr(dX), ptr(a['flag']), ptr(a['dtok']), ptr(a['dmasked']), ptr(G['vit.pos_embedding']), B, n, d,
                                      T, st), 'mask_embed_bwd')
        self._colsum(a['dmasked'], d, G['pretrain.mask_token'], B * n, d)
        self._colsum(a['dtok'], d, G['vit.to_patch_embedding.1.bias'], B * n, d)
        self._wgrad(a['dtok'], a['patches'], 'vit.to_patch_embedding.1.weight', d, self.CP, B * n)
        self._ready('embed')
        self._ready('pretrain')

    def _attn_fwd_f32(self, L, B, ph, seed):
        """f32 parity path of Attention.forward: dots = q k^T * scale (batched exact-f32 MFMA GEMM), softmax, attn v."""
        d, h, dh, N = self.d, self.h, self.dh, self.T
        qkv, S = L['qkv'], L['probs']
        sq = (N * 3 * d, dh)
        self._gemm(GEMM_NT, qkv, qkv, S, N, N, dh, 3 * d, 3 * d, N, alpha=self.scale, batch=(B, h), strideA=sq, strideB=sq,
                 strideC=(h * N * N, N * N), b_off=d)
        check(lib().ecgvit_softmax_rows(ptr(S), B * h * N, N, N, stream()), 'softmax_rows')
        Pd = S
        if ph > 0:
            Pd = self.act['pd']
            self._drop_apply_f32(S, Pd, B * h * N * N, ph, seed)
        self._gemm(GEMM_NN, Pd, qkv, L['attn'], N, dh, N, N, 3 * d, d, batch=(B, h), strideA=(h * N * N, N * N), strideB=sq,
                 strideC=(N * d, dh), b_off=2 * d)

    def _drop_apply_f32(self, src, dst, count, p, seed):
        # f32 parity path: the 16-bit pair hash of `ecgvit_dropout_apply` over element index ((b*h + head)*N + q)*N + key, exact p.
        # NOT the fused bf16 kernels' mask (one 8-bit hash per four keys, p rounded to 1/256, pitch ceil(N/4)): the two paths drop
        # different units, each consistently between its own forward and backward
        cnt8 = count // 8 * 8
        check(lib().ecgvit_dropout_apply(ptr(src), ptr(dst), cnt8, p, seed, hip.F32, stream()), 'dropout_apply')
        if cnt8 != count:
            raise ValueError('f32 attention dropout needs B*h*N*N to be a multiple of 8')

    # ---------------------------------------------------------------- backward
    def backward(self, gscalar=None, gelem=None, gscale=1.0, glogits=None, tiles_per_workgroup=0):
        """Overwrites every gradient view in gflat. Upstream: `gscalar` (1,) for the mean loss, or `gelem` (B,K) for
        reduction='none'; gscale folds the 1/(B*K) of the mean. `glogits` (B,K): extra upstream gradient on the logits.
        tiles_per_workgroup > 0: this pass's large A.B^T launches run as dispatcher-balanced chunks of about that many tiles (the
        caller overlaps RCCL collectives with the pass, whose kernels hold CUs); the setting ends with the pass, exception or not."""
        self._tpw = int(tiles_per_workgroup)
        try:
            self._backward(gscalar, gelem, gscale, glogits)
        finally:
            self._tpw = 0

    def _backward(self, gscalar, gelem, gscale, glogits):
        a, W, T = self.act, self.W, hip.code(self.dtype)
        l, st = lib(), stream()
        sv = self.saved
        B, ph, pe, seed = sv['B'], sv['ph'], sv['pe'], sv['seed']
        d, f, h, dh, N, n = self.d, self.f, self.h, self.dh, self.N, self.n
        M, Mp = B * N, B * n
        pre = 'vit.'
        G = self.G32
        if sv['labels'] is not None and (gscalar is not None or gelem is not None):
            check(l.ecgvit_bce_bwd(ptr(a['logits']), ptr(sv['labels']), ptr(sv['weight']), ptr(gscalar), ptr(gelem), gscale,
                                   ptr(a['dlogits']), B * self.K, st), 'bce_bwd')
            dlog = a['dlogits']
            if glogits is not None:
                raise NotImplementedError('simultaneous loss and logits upstream gradients')
        elif glogits is not None:
            dlog = glogits
        else:
            raise ValueError('backward needs an upstream gradient')
        dX = a['dxa']
        check(l.ecgvit_head_bwd(ptr(dlog), ptr(a['xhat']), ptr(a['hrstd']), ptr(self.P32[pre + 'mlp_head.0.weight']),
                                ptr(self.P32[pre + 'mlp_head.0.bias']), ptr(self.P32[pre + 'mlp_head.1.weight']),
                                ptr(G[pre + 'mlp_head.1.weight']), ptr(G[pre + 'mlp_head.1.bias']),
                                ptr(G[pre + 'mlp_head.0.weight']), ptr(G[pre + 'mlp_head.0.bias']), ptr(dX), N, B, d, self.K,
                                T, st), 'head_bwd')
        self._ready('head')
        dX = self._trunk_bwd(dX, a['dxb'])
        for k in G:
            if k.startswith('pretrain.'):
                G[k].zero_()   # the masked-objective head takes no part in the supervised step
        self._ready('pretrain')
        # ---- embedding backward
        check(l.ecgvit_embed_bwd(ptr(dX), ptr(a['dtok']), ptr(G[pre + 'cls_token']), ptr(G[pre + 'pos_embedding']), B, n, d, pe,
                                 seed + 1, T, st), 'embed_bwd')
        self._colsum(a['dtok'], d, G[pre + 'to_patch_embedding.1.bias'], Mp, d)
        self._wgrad(a['dtok'], a['patches'], pre + 'to_patch_embedding.1.weight', d, self.CP, Mp)
        self._ready('embed')

    def _ready(self, tag):
        if self.on_grads_ready is not None:
            self.on_grads_ready(tag)

    def _trunk_bwd(self, dX, other):
        """backward of _trunk_fwd: consumes dX = d(loss)/d(x_L) ([B*T, d]), fills every layer's parameter gradients, returns d(x_0)"""
        a, W, T = self.act, self.W, hip.code(self.dtype)
        l, st = lib(), stream()
        sv = self.saved
        B, ph, seed = sv['B'], sv['ph'], sv['seed']
        d, f, h, dh, N = self.d, self.f, self.h, self.dh, self.T
        M = B * N
        pre = 'vit.'
        G = self.G32
        # dY = gradient entering the current `dropout(Linear + bias) + residual` site (masked copy of dX when dropout is on);
        # after the first site, the fused LayerNorm backward of the previous stage has already produced it AND its bias gradient
        have = False
        dY = dX
        pq4 = pq6 = False   # fp8_linear: the LayerNorm backward before a site already wrote its e5m2 operand copy
        for i in reversed(range(self.Ly)):
            L = a['layers'][i]
            lp = f'{pre}transformer.layers.{i}.'
            s0 = seed + 100 * (i + 1)
            Xin = a['x0'] if i == 0 else a['layers'][i - 1]['x2']
            # ---- FeedForward backward: x2 = drop(hact W2^T + b2) + x1
            if not have:
                dY = dX
                if ph > 0:
                    self._drop_apply(dX, a['dxm'], M * d, ph, s0 + 4)
                    dY = a['dxm']
                self._colsum(dY, d, G[lp + '1.fn.net.3.bias'], M, d)
            f8 = self.fp8 and M >= 2048
            g4 = self._grad8(8 * i + 4, dY, M * d, prequant='q8' if pq4 else False) if f8 else None
            self._wgrad(dY, L['hact'], lp + '1.fn.net.3.weight', d, f, M, pre=g4, x8=L.get('hact_8'), xsite=8 * i + 3)
            # dgrad with GELU' (+ dropout mask) epilogue; the epilogue also reduces the columns = gradient of the FFN-up bias
            if self.dtype == torch.bfloat16:
                epi, pdrop = EPI_MUL_AUX | EPI_COLSUM | (hip.EPI_AUX8 if self._aux8(M) else 0), 0.0
            else:
                epi, pdrop = EPI_GELU_BWD | EPI_COLSUM | (EPI_DROPOUT if ph > 0 else 0), ph
            dq = self._dgrad(dY, lp + '1.fn.net.3.weight', a['dh'], M, f, d, site=8 * i + 4, emit_site=8 * i + 5, pre=g4, emit_only8=self._only8(M),
                             epilogue=epi, aux=L['hpre'],
                             ldaux=f, dropout_p=pdrop, seed=s0 + 3, workspace=a['ws'], colsum_out=G[lp + '1.fn.net.0.bias'])
            g5 = self._grad8(8 * i + 5, a['dh'], M * f, prequant=bool(dq)) if f8 else None
            self._wgrad(a['dh'], L['xn2'], lp + '1.fn.net.0.weight', f, d, M, pre=g5, x8=L.get('xn2_8'), xsite=8 * i + 2)
            self._dgrad(a['dh'], lp + '1.fn.net.0.weight', a['dxn'], M, d, f, site=8 * i + 5, pre=g5)
            # LN2 backward; its output feeds the attention out-projection site (mask seed s0+2, bias to_out.0.bias)
            pq6 = self._ln_bwd_fused(a['dxn'], L['x1'], self.P32[lp + '1.norm.weight'], L['mean2'], L['rstd2'], dX, other,
                                     G[lp + '1.norm.weight'], G[lp + '1.norm.bias'], M, a['dxm'], G[lp + '0.fn.to_out.0.bias'], ph, s0 + 2,
                                     q8_site=8 * i + 6)
            dX, other = other, dX  # dX = d(x1)
            dY = a['dxm'] if ph > 0 else dX
            # ---- Attention backward: x1 = drop(attn Wo^T + bo) + x
            g6 = self._grad8(8 * i + 6, dY, M * d, prequant='q8' if pq6 else False) if f8 else None
            self._wgrad(dY, L['attn'], lp + '0.fn.to_out.0.weight', d, d, M, pre=g6, x8=L.get('attn_8'), xsite=8 * i + 1)
            self._dgrad(dY, lp + '0.fn.to_out.0.weight', a['dattn'], M, d, d, site=8 * i + 6, pre=g6)
            pq7 = False   # fp8_linear: the attention backward wrote the e5m2 copy of dqkv itself (into the operand scratch)
            if self.dtype == torch.bfloat16:
                if f8 and (8 * i + 7) in self._f8_seen and 128 < N <= 512 and N * 3 * d * 2 < 2 ** 31:
                    check(l.ecgvit_attention_bwd_q8(ptr(L['qkv']), ptr(L['attn']), ptr(a['dattn']), ptr(L['lse']), ptr(a['dqkv']), B, N, h, dh, self.scale,
                                                    ph, s0 + 1, ptr(a['q8']), ptr(self.f8_scale[8 * i + 7:8 * i + 8]), ptr(self.f8_amax[8 * i + 7:8 * i + 8]), st),
                          'attention_bwd_q8')
                    pq7 = True
                else:
                    check(l.ecgvit_attention_bwd(ptr(L['qkv']), ptr(L['attn']), ptr(a['dattn']), ptr(L['lse']), ptr(a['dqkv']), B, N, h,
                                                 dh, self.scale, ph, s0 + 1, T, st), 'attention_bwd')
            else:
                self._attn_bwd_f32(L, B, ph, s0 + 1)
            g7 = self._grad8(8 * i + 7, a['dqkv'], M * 3 * d, prequant='q8' if pq7 else False) if f8 else None
            self._wgrad(a['dqkv'], L['xn1'], lp + '0.fn.to_qkv.weight', 3 * d, d, M, pre=g7, x8=L.get('xn1_8'), xsite=8 * i)
            self._dgrad(a['dqkv'], lp + '0.fn.to_qkv.weight', a['dxn'], M, d, 3 * d, site=8 * i + 7, pre=g7)
            if i > 0:
                # LN1 backward; its output feeds layer i-1's FFN-down site (mask seed of layer i-1, bias net.3.bias)
                lq = f'{pre}transformer.layers.{i - 1}.'
                pq4 = self._ln_bwd_fused(a['dxn'], Xin, self.P32[lp + '0.norm.weight'], L['mean1'], L['rstd1'], dX, other,
                                         G[lp + '0.norm.weight'], G[lp + '0.norm.bias'], M, a['dxm'], G[lq + '1.fn.net.3.bias'], ph,
                                         seed + 100 * i + 4, q8_site=8 * (i - 1) + 4)
                have = True
            else:
                self._ln_bwd(a['dxn'], Xin, self.P32[lp + '0.norm.weight'], L['mean1'], L['rstd1'], dX, other,
                             G[lp + '0.norm.weight'], G[lp + '0.norm.bias'], M)
            dX, other = other, dX
            dY = a['dxm'] if ph > 0 else dX
            # every gradient of layer i is final here (its net.3.bias came from the fused LN1 backward of layer i+1, earlier)
            self._ready(f'layer{i}')
        return dX

    def _attn_bwd_f32(self, L, B, ph, seed):
        d, h, dh, N = self.d, self.h, self.dh, self.T
        a = self.act
        qkv, P, dqkv, dO, dP = L['qkv'], L['probs'], a['dqkv'], a['dattn'], a['dp']
        sq, so, sp = (N * 3 * d, dh), (N * d, dh), (h * N * N, N * N)
        Pd = P
        if ph > 0:
            Pd = a['pd']
            self._drop_apply_f32(P, Pd, B * h * N * N, ph, seed)
        # dV = Pd^T dO
        self._gemm(GEMM_TN, Pd, dO, dqkv, N, dh, N, N, d, 3 * d, batch=(B, h), strideA=sp, strideB=so, strideC=sq, c_off=2 * d)
        # dPd = dO V^T
        self._gemm(GEMM_NT, dO, qkv, dP, N, N, dh, d, 3 * d, N, batch=(B, h), strideA=so, strideB=sq, strideC=sp, b_off=2 * d)
        if ph > 0:
            self._drop_apply_f32(dP, dP, B * h * N * N, ph, seed)
        # dS = P * (dP - rowsum(P dP)) * scale
        check(lib().ecgvit_softmax_bwd_rows(ptr(P), ptr(dP), B * h * N, N, N, self.scale, stream()), 'softmax_bwd_rows')
        # dQ = dS K ; dK = dS^T Q
        self._gemm(GEMM_NN, dP, qkv, dqkv, N, dh, N, N, 3 * d, 3 * d, batch=(B, h), strideA=sp, strideB=sq, strideC=sq, b_off=d)
        self._gemm(GEMM_TN, dP, qkv, dqkv, N, dh, N, N, 3 * d, 3 * d, batch=(B, h), strideA=sp, strideB=sq, strideC=sq, c_off=d)

    # ---------------------------------------------------------------- per-layer attention probabilities (f3)
    def attention_probs(self, layer):
        """Post-softmax attention of `layer` for the last forward, (B, h, N, N) f32 -- what vit_pytorch's Recorder hooks
        (reference ecg_vit.py:176-194). The f32 path keeps them; the fused bf16 path rebuilds them from its saved qkv + log-sum-exp."""
        B, L = self.saved['B'], self.act['layers'][layer]
        if self.dtype == torch.float32:
            return L['probs'].view(B, self.h, self.T, self.T)
        out = torch.empty(B, self.h, self.T, self.T, dtype=torch.float32, device=L['qkv'].device)
        check(lib().ecgvit_attention_probs(ptr(L['qkv']), ptr(L['lse']), ptr(out), B, self.T, self.h, self.dh, self.scale, hip.BF16, stream()),
              'attention_probs')
        return out
