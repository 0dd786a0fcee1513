"""
Generate tests/golden/*.npz + *.json by EXECUTING THE REFERENCE'S OWN FILES (TEST INFRASTRUCTURE).

Runs only in the build container (needs /root/reference); the fixtures it writes are data
(inputs + expected outputs) and are committed, this script with them.  Nothing here, and
nothing under /root/reference, is read by tests / smoke / bench at run time.

What is real and what is a stand-in:
  real      ecg_transformer.models.ecg_vit (EcgVitConfig, EcgVit), ecg_transformer.models.train
            (get_train_args), ecg_transformer.util (log_dict_p, ca, config), util.models.ModelOutput,
            transformers' PretrainedConfig + get_{cosine,constant}_schedule_with_warmup,
            torch BCEWithLogitsLoss / AdamW / clip_grad_norm_, preprocess.transform (TimeEndPad...)
  stubbed   UI/IO-only imports that are not installed here (sty, colorama, seaborn, h5py, wfdb,
            torchvision, pytorch_lightning, tensorboard, icecream, loess) -- none touches a number
  stand-in  `vit_pytorch.ViT` := oracle.vit_oracle.OracleViT  (vit-pytorch==0.33.2 is not vendored;
            its arithmetic is therefore **parity unpinned**, see oracle/__init__.py)

usage:  python oracle/make_golden.py            (from the repo root)
"""
import os
import sys
import copy
import json
import types
import importlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)


def _install_stubs():
    import transformers  # noqa: F401  real; import before stubbing anything it might probe

    def sink(k):
        if k.startswith('__'):
            raise AttributeError(k)
        return _Any()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        """attribute/call sink for styling objects used at class-definition time"""
        def __init__(self, *a, **k): pass
        def __getattr__(self, k): return _Any()
        def __call__(self, *a, **k): return _Any()
        def __add__(self, o): return o if isinstance(o, str) else self
        def __radd__(self, o): return o if isinstance(o, str) else self
        def __str__(self): return ''
        def __format__(self, spec): return ''

    mod('sty', fg=_Any(), bg=_Any(), ef=_Any(), rs=_Any(), Style=_Any, RgbFg=_Any)
    mod('colorama', Fore=_Any(), Style=_Any(), init=lambda *a, **k: None)
    mod('seaborn', set_style=lambda *a, **k: None, set_context=lambda *a, **k: None,
        color_palette=lambda *a, **k: [], __getattr__=sink)
    mod('h5py', File=_Any)
    wf = mod('wfdb', rdrecord=_Any, rdsamp=_Any)
    wf.processing = mod('wfdb.processing', resample_sig=_Any)
    mod('icecream', ic=lambda *a, **k: None)
    mod('loess', __getattr__=sink)
    mod('loess.loess_1d', loess_1d=_Any)
    tv = mod('torchvision')
    tv.transforms = mod('torchvision.transforms', Compose=lambda ts: (lambda x: [x := t(x) for t in ts][-1]))
    pl = mod('pytorch_lightning', LightningModule=torch.nn.Module, LightningDataModule=object,
             Trainer=_Any, seed_everything=lambda s: torch.manual_seed(s))
    cb = mod('pytorch_lightning.callbacks', ModelCheckpoint=_Any, EarlyStopping=_Any, LearningRateMonitor=_Any)
    cb.progress = mod('pytorch_lightning.callbacks.progress', ProgressBar=object, TQDMProgressBar=object,
                      ProgressBarBase=object)
    pl.callbacks = cb
    pl.loggers = mod('pytorch_lightning.loggers', TensorBoardLogger=_Any)
    tb = mod('torch.utils.tensorboard', SummaryWriter=_Any)
    torch.utils.tensorboard = tb

    from oracle.vit_oracle import OracleViT
    vp = mod('vit_pytorch', ViT=OracleViT)
    vp.recorder = mod('vit_pytorch.recorder', Recorder=_Any)


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: later in-place optimiser steps must not alias saved arrays


def metrics_golden():
    """(12) eval metrics ("next" row f1): the reference's own `get_accuracy` (util/train.py:12-56, sklearn underneath) run on
    seeded probabilities / multi-hot labels: ties, absent classes, the all-rows-equal case (no AUROC), a single valid class"""
    ref_train_util = importlib.import_module('ecg_transformer.util.train')
    rng = np.random.default_rng(77)
    cases = {}

    def case(name, probs, labels):
        out = ref_train_util.get_accuracy(torch.from_numpy(probs), torch.from_numpy(labels), return_auc=True)
        cases[name] = dict(
            probs=probs.tolist(), labels=labels.tolist(),
            expect={k: (None if v is None else ({kk: float(vv) for kk, vv in v.items()} if isinstance(v, dict) else float(v)))
                    for k, v in out.items()})
    K = 71
    prior = np.concatenate([np.full(8, 0.3), np.full(23, 0.05), np.full(40, 0.004)])        # PTB-XL-like: most codes are rare
    lb = (rng.random((96, K)) < prior).astype(np.float32)
    pr = np.clip(0.35 * lb + rng.random((96, K)) * 0.7, 0, 1).astype(np.float32)
    case('b96_rare', pr, lb)
    case('b96_ties', (np.round(pr * 8) / 8).astype(np.float32), lb)                          # heavy ties incl. exactly 0.5
    lb1 = np.tile(lb[:1], (8, 1))
    case('b8_rows_equal', pr[:8].copy(), lb1)                                                 # msk_2_class empty: macro_auc None
    lb2 = lb1.copy(); lb2[3, 5] = 1 - lb2[3, 5]
    case('b8_one_class', pr[:8].copy(), lb2)                                                  # roc_auc_score returns a scalar
    case('b4_all_neg_pred', np.full((4, K), 0.1, np.float32), lb[:4].copy())                  # zero_division branches
    with open(os.path.join(OUT, 'metrics.json'), 'w') as f:
        json.dump(dict(id2code=list(ref_train_util.get_accuracy.id2code), cases=cases), f)
    print('metrics:', {k: (v['expect']['binary_accuracy'], v['expect']['macro_auc']) for k, v in cases.items()})


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    if '--only-metrics' in sys.argv:
        return metrics_golden()
    ecg_vit = importlib.import_module('ecg_transformer.models.ecg_vit')
    ref_train = importlib.import_module('ecg_transformer.models.train')
    ref_util = importlib.import_module('ecg_transformer.util')
    transform = importlib.import_module('ecg_transformer.preprocess.transform')
    from transformers import get_cosine_schedule_with_warmup, get_constant_schedule_with_warmup
    EcgVitConfig, EcgVit = ecg_vit.EcgVitConfig, ecg_vit.EcgVit
    os.makedirs(OUT, exist_ok=True)
    seed = ref_util.config('random-seed')
    assert seed == 77

    # ---- (8)(9) from_defined field table, param counts, meta strings -------------------------
    table = {}
    for nm in ref_util.ca.model_names if hasattr(ref_util.ca, 'model_names') else []:
        conf = EcgVitConfig.from_defined(nm)
        m = EcgVit(config=conf)
        table[nm] = dict(
            size=conf.size, hidden_size=conf.hidden_size, num_hidden_layers=conf.num_hidden_layers,
            num_attention_heads=conf.num_attention_heads, intermediate_size=conf.intermediate_size,
            max_signal_length=conf.max_signal_length, patch_size=conf.patch_size, num_channels=conf.num_channels,
            hidden_dropout_prob=conf.hidden_dropout_prob,
            attention_probs_dropout_prob=conf.attention_probs_dropout_prob, num_class=conf.num_class,
            n_param=sum(p.numel() for p in m.parameters()),
            meta=m.meta, meta_str=m.meta_str, to_str=m.to_str(),
            state_dict_keys=[[k, list(v.shape)] for k, v in m.state_dict().items()] if nm in (
                'ecg-vit-debug',) else None,
        )
        del m
    default_conf = EcgVitConfig()
    table['__default__'] = {k: getattr(default_conf, k) for k in (
        'max_signal_length', 'patch_size', 'num_channels', 'hidden_size', 'num_hidden_layers',
        'num_attention_heads', 'intermediate_size', 'hidden_dropout_prob', 'attention_probs_dropout_prob',
        'num_class', 'size')}
    errs = {}
    for bad, kw in (('model_name', dict(model_name='ecg-vit-huge')), ('optimizer', dict(optimizer='SGD')),
                    ('schedule', dict(schedule='linear'))):
        try:
            ref_util.ca(**kw)
            errs[bad] = None
        except Exception as e:  # noqa
            errs[bad] = type(e).__name__
    try:
        EcgVit(config=EcgVitConfig(hidden_size=30, num_attention_heads=4))
        errs['d_mod_h'] = None
    except Exception as e:  # noqa
        errs['d_mod_h'] = type(e).__name__
    try:
        EcgVit(config=EcgVitConfig(max_signal_length=2500, patch_size=64))
        errs['l_mod_p'] = None
    except Exception as e:  # noqa
        errs['l_mod_p'] = type(e).__name__
    table['__errors__'] = errs

    # ---- (10) get_train_args incl. the floor-division quirk; (7) LR vectors -------------------
    tr_args = []
    for n_train, bsz, ep, extra in ((17441, 256, 32, dict(warmup_ratio=0.1)), (17441, 64, 3, {}), (1000, 64, 3, {}),
                                    (128, 64, 2, dict(schedule='constant')), (None, 64, 3, {})):
        a = ref_train.get_train_args(dict(train_batch_size=bsz, num_train_epoch=ep, **extra), n_train=n_train)
        a = {k: v for k, v in a.items() if k != 'precision'}
        tr_args.append(dict(n_train=n_train, args=a))
    lrs = {}
    for tag, (sch, n_step, ratio, lr) in dict(cos_2176=('cosine', 2176, 0.1, 3e-4), cos_30=('cosine', 30, 0.05, 3e-4),
                                               const_20=('constant', 20, 0.25, 1e-3)).items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr)
        n_w = round(n_step * ratio)
        s = (get_constant_schedule_with_warmup(opt, num_warmup_steps=n_w) if sch == 'constant'
             else get_cosine_schedule_with_warmup(opt, num_warmup_steps=n_w, num_training_steps=n_step))
        v = [s.get_last_lr()[0]]
        for _ in range(n_step):
            opt.step()
            s.step()
            v.append(s.get_last_lr()[0])
        lrs[tag] = dict(schedule=sch, n_step=n_step, warmup_ratio=ratio, lr=lr, n_warmup=n_w, values=v)
    with open(os.path.join(OUT, 'host_contract.json'), 'w') as f:
        json.dump(dict(from_defined=table, train_args=tr_args, lr=lrs), f, indent=1, default=str)

    # ---- (1) patch gather on an arange input: bit-exact integer pin ---------------------------
    gather = {}
    for (L, P) in ((2560, 64), (5000, 20), (40, 10)):
        conf = EcgVitConfig(max_signal_length=L, patch_size=P, hidden_size=8, num_hidden_layers=1,
                            num_attention_heads=2, intermediate_size=8)
        m = EcgVit(config=conf)
        x = torch.arange(2 * 12 * L, dtype=torch.float32).reshape(2, 12, L)
        tok = m.vit.to_patch_embedding[0](x.unsqueeze(-2))
        gather[f'L{L}_P{P}'] = _np(tok).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, 'patch_gather.npz'), **gather)

    # ---- (2)-(6) micro-config forward / intermediates / grads / AdamW steps -------------------
    micro = dict(
        # fp32-path micro config at the reference default geometry and at the benchmark geometry
        g2560=dict(max_signal_length=2560, patch_size=64, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=64, B=3),
        g5000=dict(max_signal_length=5000, patch_size=20, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                   intermediate_size=64, B=2),
        # dh=64 config (BASELINE configs[0] "tiny 2-layer d=128"), exercised by the bf16 MFMA path too
        t128=dict(max_signal_length=1000, patch_size=20, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                  intermediate_size=256, B=2),
    )
    for tag, spec in micro.items():
        spec = dict(spec)
        B = spec.pop('B')
        torch.manual_seed(seed)
        conf = EcgVitConfig(hidden_dropout_prob=0., attention_probs_dropout_prob=0., **spec)
        model = EcgVit(config=conf)
        model.train()
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, 12, conf.max_signal_length, generator=g)
        y = (torch.rand(B, 71, generator=g) < 0.04).float()
        y[0, 3] = 1.
        blob = {f'param/{k}': _np(v) for k, v in model.state_dict().items()}
        blob['cfg'] = np.frombuffer(json.dumps(spec).encode(), dtype=np.uint8)
        blob['x'], blob['y'] = _np(x), _np(y)

        inter = {}

        def hook(name):
            def fn(_m, _i, o):
                inter[name] = _np(o)
            return fn
        hs = [model.vit.to_patch_embedding.register_forward_hook(hook('embed'))]
        for i, (attn, ff) in enumerate(model.vit.transformer.layers):
            hs += [attn.norm.register_forward_hook(hook(f'l{i}/ln1')),
                   attn.fn.to_qkv.register_forward_hook(hook(f'l{i}/qkv')),
                   *([attn.fn.attend.register_forward_hook(hook(f'l{i}/probs'))] if i == 0 else []),
                   attn.register_forward_hook(hook(f'l{i}/attn_out')),
                   ff.norm.register_forward_hook(hook(f'l{i}/ln2')),
                   ff.fn.net[1].register_forward_hook(hook(f'l{i}/gelu')),
                   ff.register_forward_hook(hook(f'l{i}/ff_out'))]
        hs.append(model.vit.transformer.register_forward_hook(hook('trunk')))
        out = model(sample_values=x, labels=y)
        for h in hs:
            h.remove()
        blob.update({f'inter/{k}': v for k, v in inter.items()})
        blob['logits'], blob['loss_mean'] = _np(out.logits), _np(out.loss)
        assert model(sample_values=x).loss is None
        model.loss_reduction = 'none'
        blob['loss_none'] = _np(model(sample_values=x, labels=y).loss)
        model.loss_reduction = 'mean'
        # on a COPY: the reference's forward replaces `self.loss_fn` by a weighted BCE object (ecg_vit.py:147) that
        # stays installed after `loss_weight` is reset to None -- a statefulness quirk the fixtures must not inherit
        wm = copy.deepcopy(model)
        wm.loss_weight = [1.0, 3.0]
        blob['loss_weighted'] = _np(wm(sample_values=x, labels=y).loss)
        del wm
        model.eval()
        blob['logits_eval'] = _np(model(sample_values=x).logits)
        model.train()

        # train steps exactly as train.py:241-252 / :271-283 wires them
        n_step = 6
        opt = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=1e-2)
        sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=round(n_step * 0.34), num_training_steps=n_step)
        blob['train/n_step'], blob['train/warmup_ratio'] = np.int64(n_step), np.float64(0.34)
        losses, norms = [], []
        for it in range(3):
            opt.zero_grad()
            o = model(sample_values=x, labels=y)
            o.loss.backward()
            if it == 0:
                blob.update({f'grad/{k}': _np(p.grad) for k, p in model.named_parameters()})
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0, error_if_nonfinite=True)
            opt.step()
            sch.step()
            losses.append(float(o.loss))
            norms.append(float(tn))
            if it in (0, 2):
                blob.update({f'param_after{it + 1}/{k}': _np(v) for k, v in model.state_dict().items()})
        blob['train/losses'], blob['train/grad_norms'] = np.array(losses, np.float64), np.array(norms, np.float64)
        np.savez_compressed(os.path.join(OUT, f'micro_{tag}.npz'), **blob)
        print(tag, 'loss', float(blob['loss_mean']), 'grad-norm', norms, 'bytes',
              os.path.getsize(os.path.join(OUT, f'micro_{tag}.npz')))

    # ---- (11) TimeEndPad incl. the L % P == 0 quirk, Normalize ("next" row f2) ----------------
    sig = np.arange(2 * 12 * 50, dtype=np.float32).reshape(2, 12, 50)
    pads = {}
    for k in (20, 25, 64):
        pads[f'pad_k{k}'] = transform.TimeEndPad(k, pad_kwargs=dict(mode='constant', constant_values=0))(sig)
    stats = ref_util.config('datasets.PTB-XL.train-stats.denoised')
    pads['norm_mean'], pads['norm_std'] = np.asarray(stats['mean'], np.float32), np.asarray(stats['std'], np.float32)
    pads['norm_out'] = transform.Normalize(**stats)(sig)
    pads['sig'] = sig
    np.savez_compressed(os.path.join(OUT, 'transforms.npz'), **pads)
    metrics_golden()
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
