"""-m gpu: the whole HIP path (EcgVit.forward / loss.backward / fused train step) against
  (1) the committed golden fixtures (outputs of the reference's own wrapper code around the oracle ViT) and
  (2) the CPU oracle on the same seeded inputs,
f32 path within 1e-4 relative (north_star tolerance), bf16 path within bf16 rounding (stated per assert).
Size-independent properties at the benchmark geometry: batch-slice invariance, run-to-run determinism."""
import os

import numpy as np
import pytest
import torch

from conftest import Cfg, load_micro
from hiputil import rel_err, max_err
from oracle import vit_oracle as O
import ecg_representation_learning_amd as E

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16


def build(tag, dtype, **over):
    z, spec = load_micro(tag)
    conf = E.EcgVitConfig(hidden_dropout_prob=0., attention_probs_dropout_prob=0., **{**spec, **over})
    m = E.EcgVit(config=conf, compute_dtype=dtype)
    sd = {k[len('param/'):]: torch.from_numpy(z[k]) for k in z.files if k.startswith('param/')}
    m.load_state_dict(sd, strict=True)
    return z, m.cuda()


@pytest.mark.parametrize('tag', ['g2560', 'g5000', 't128'])
def test_f32_forward_matches_golden(tag):
    z, m = build(tag, F32)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    out = m(sample_values=x, labels=y)
    assert isinstance(out, E.ModelOutput) and out.logits.shape == (x.shape[0], 71) and out.loss.ndim == 0
    assert abs(float(out.loss) - float(z['loss_mean'])) / float(z['loss_mean']) < 1e-5
    assert max_err(out.logits, torch.from_numpy(z['logits'])) < 1e-4 * max(1.0, float(np.abs(z['logits']).max()))
    # intermediates straight out of the engine's activation slabs
    act = m._engine().act
    B = x.shape[0]
    for i, L in enumerate(act['layers']):
        assert rel_err(L['xn1'], torch.from_numpy(z[f'inter/l{i}/ln1']).reshape(L['xn1'].shape)) < 1e-5
        assert rel_err(L['qkv'], torch.from_numpy(z[f'inter/l{i}/qkv']).reshape(L['qkv'].shape)) < 1e-5
        assert rel_err(L['hact'], torch.from_numpy(z[f'inter/l{i}/gelu']).reshape(L['hact'].shape)) < 1e-5
    assert rel_err(act['layers'][-1]['x2'], torch.from_numpy(z['inter/trunk']).reshape(act['layers'][-1]['x2'].shape)) < 1e-5
    assert rel_err(m.attention_probs(0), torch.from_numpy(z['inter/l0/probs'])) < 1e-5
    assert m(sample_values=x).loss is None
    m.loss_reduction = 'none'
    ln = m(sample_values=x, labels=y).loss
    assert ln.shape == (B, 71) and rel_err(ln, torch.from_numpy(z['loss_none'])) < 1e-5
    m.loss_reduction = 'mean'
    m.loss_weight = [1.0, 3.0]
    assert abs(float(m(sample_values=x, labels=y).loss) - float(z['loss_weighted'])) / float(z['loss_weighted']) < 1e-5
    m.loss_weight = None
    m.eval()
    with torch.no_grad():
        assert max_err(m(sample_values=x).logits, torch.from_numpy(z['logits_eval'])) < 1e-4


@pytest.mark.parametrize('tag', ['g2560', 'g5000', 't128'])
def test_f32_backward_matches_golden(tag):
    z, m = build(tag, F32)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    out = m(sample_values=x, labels=y)
    out.loss.backward()
    gn = 0.0
    for k, p in m.named_parameters():
        ref = torch.from_numpy(z[f'grad/{k}'])
        assert p.grad is not None, k
        assert rel_err(p.grad, ref) < 1e-4, (k, rel_err(p.grad, ref))
        gn += float(p.grad.double().pow(2).sum())
    assert abs(gn ** 0.5 - z['train/grad_norms'][0]) / z['train/grad_norms'][0] < 1e-5
    # loss 'none' with an explicit upstream gradient == mean loss gradient
    m.zero_grad()
    m.loss_reduction = 'none'
    ln = m(sample_values=x, labels=y).loss
    ln.mean().backward()
    k = 'vit.transformer.layers.0.0.fn.to_qkv.weight'
    assert rel_err(dict(m.named_parameters())[k].grad, torch.from_numpy(z[f'grad/{k}'])) < 1e-4


@pytest.mark.parametrize('tag', ['g2560', 't128'])
def test_f32_fused_train_steps_match_golden(tag):
    z, m = build(tag, F32)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    args = E.get_train_args(dict(learning_rate=3e-4, weight_decay=1e-2, warmup_ratio=float(z['train/warmup_ratio'])))
    args['n_step'] = int(z['train/n_step'])
    ts = E.HipTrainStep(m, args, sync_nonfinite=True)
    for it in range(3):
        loss, _ = ts.step(x, y)
        assert abs(float(loss) - z['train/losses'][it]) / z['train/losses'][it] < 1e-5
        assert abs(ts.grad_norm() - z['train/grad_norms'][it]) / z['train/grad_norms'][it] < 1e-4
        if it in (0, 2):
            for k, v in m.state_dict().items():
                assert max_err(v, torch.from_numpy(z[f'param_after{it + 1}/{k}'])) < 3e-6, (it, k)


def test_f32_torch_optimizer_interop_matches_golden():
    """the reference's own step body verbatim: zero_grad / forward / backward / clip_grad_norm_ / AdamW / scheduler"""
    z, m = build('g2560', F32)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=1e-2)
    n_step = int(z['train/n_step'])
    sch = torch.optim.lr_scheduler.LambdaLR(opt, E.lr_multiplier('cosine', round(n_step * float(z['train/warmup_ratio'])), n_step))
    for it in range(3):
        opt.zero_grad()
        out = m(sample_values=x, labels=y)
        out.loss.backward()
        tn = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0, error_if_nonfinite=True)
        opt.step()
        sch.step()
        assert abs(float(tn) - z['train/grad_norms'][it]) / z['train/grad_norms'][it] < 1e-4
    for k, v in m.state_dict().items():
        assert max_err(v, torch.from_numpy(z[f'param_after3/{k}'])) < 3e-6, k
    # flat-buffer clip helper gives the same norm
    opt.zero_grad()
    m(sample_values=x, labels=y).loss.backward()
    n1 = float(E.clip_grad_norm_(m, 1.0))
    g2 = sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()) ** 0.5
    assert n1 > 0 and g2 <= 1.0 + 1e-4


def test_bf16_matches_golden_t128():
    z, m = build('t128', BF16)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    out = m(sample_values=x, labels=y)
    rel = abs(float(out.loss) - float(z['loss_mean'])) / float(z['loss_mean'])
    assert rel < 2e-2, rel                                   # bf16 activations: ~1e-2 relative on the loss
    assert max_err(out.logits, torch.from_numpy(z['logits'])) < 0.1
    out.loss.backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(z[f'grad/{k}'])
        cos = float((p.grad.cpu().double().flatten() @ ref.double().flatten()) / (p.grad.cpu().double().norm() * ref.double().norm()))
        assert cos > 0.98, (k, cos)
        assert rel_err(p.grad, ref) < 0.15, (k, rel_err(p.grad, ref))
    args = E.get_train_args(dict(learning_rate=3e-4, weight_decay=1e-2, warmup_ratio=float(z['train/warmup_ratio'])))
    args['n_step'] = int(z['train/n_step'])
    ts = E.HipTrainStep(m, args, sync_nonfinite=True)
    for it in range(3):
        loss, _ = ts.step(x, y)
        assert abs(float(loss) - z['train/losses'][it]) / z['train/losses'][it] < 3e-2


def _oracle_pair(conf_kw, B, dtype, seed=77):
    conf = E.EcgVitConfig(hidden_dropout_prob=0., attention_probs_dropout_prob=0., **conf_kw)
    torch.manual_seed(seed)
    ref = O.OracleEcgVit(config=conf)
    m = E.EcgVit(config=conf, compute_dtype=dtype)
    m.load_state_dict(ref.state_dict())
    x, y = O.synthetic_batch(B, length=conf.max_signal_length, seed=seed)
    return conf, ref, m.cuda(), x, y


def test_f32_vs_oracle_benchmark_geometry():
    """12 x 5000, patch 20 (251 tokens), reference 'tiny' (4 layers, d=256, 4 heads): forward/loss/grads vs the CPU oracle"""
    kw = dict(max_signal_length=5000, patch_size=20, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024)
    conf, ref, m, x, y = _oracle_pair(kw, 8, F32)
    ref.train(); m.train()
    o_ref = ref(sample_values=x, labels=y)
    o_ref.loss.backward()
    out = m(sample_values=x.cuda(), labels=y.cuda())
    out.loss.backward()
    assert abs(float(out.loss) - float(o_ref.loss)) / float(o_ref.loss) < 1e-4
    assert max_err(out.logits, o_ref.logits) < 1e-4
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad, q.grad) < 1e-4, (k, rel_err(p.grad, q.grad))


def test_bf16_vs_f32_base_geometry_and_invariants():
    """EcgVit-base layer shape (d=768, 12 heads, ffn 3072; 3 layers to bound test time) at 12x5000 / patch 20:
    bf16 vs f32 HIP paths agree within bf16 noise; eval logits are independent of the batch a record sits in; reruns are bit-identical."""
    kw = dict(max_signal_length=5000, patch_size=20, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072)
    conf, ref, m16, x, y = _oracle_pair(kw, 24, BF16)
    m32 = E.EcgVit(config=conf, compute_dtype=F32)
    m32.load_state_dict(ref.state_dict())
    m32.cuda().train(); m16.train()
    xc, yc = x.cuda(), y.cuda()
    o32 = m32(sample_values=xc, labels=yc)
    o16 = m16(sample_values=xc, labels=yc)
    assert abs(float(o16.loss) - float(o32.loss)) / float(o32.loss) < 2e-2
    o32.loss.backward(); o16.loss.backward()
    tot32 = torch.cat([p.grad.flatten() for p in m32.parameters()])
    tot16 = torch.cat([p.grad.flatten() for p in m16.parameters()])
    cos = float((tot32.double() @ tot16.double()) / (tot32.double().norm() * tot16.double().norm()))
    assert cos > 0.99, cos
    m16.eval()
    with torch.no_grad():
        la = m16(sample_values=xc).logits.clone()
        lb = m16(sample_values=xc).logits.clone()
        lc = m16(sample_values=xc[5:13].contiguous()).logits.clone()
    assert torch.equal(la, lb)                      # determinism
    assert torch.equal(la[5:13], lc)                # batch-slice invariance (records are independent; tiles are deterministic)


def test_dropout_training_path_is_consistent_fd():
    """dropout > 0 (f32 path, fixed seed): the backward applies exactly the forward's masks -- checked by a directional
    finite difference of the loss through the engine with the seed pinned."""
    kw = dict(max_signal_length=400, patch_size=20, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128)
    conf = E.EcgVitConfig(hidden_dropout_prob=0.2, attention_probs_dropout_prob=0.1, **kw)
    torch.manual_seed(3)
    m = E.EcgVit(config=conf, compute_dtype=F32).cuda().train()
    x, y = O.synthetic_batch(8, length=400, seed=5)   # 8*4*21*21 % 8 == 0
    x, y = x.cuda(), y.cuda()
    eng = m._engine()
    seed = 4242
    one = torch.ones(1, device='cuda')

    def loss_at():
        _, _, lm = eng.forward(x, y, None, training=True, seed=seed, want_mean=True)
        return float(lm)
    l0 = loss_at()
    eng.backward(gscalar=one, gscale=1.0 / (8 * 71))
    g = m._gflat.clone()
    # eval-mode loss differs (dropout active in training)
    _, _, le = eng.forward(x, y, None, training=False, seed=seed, want_mean=True)
    assert abs(float(le) - l0) > 1e-5
    torch.manual_seed(0)
    v = torch.randn_like(m._pflat)
    v = v / v.norm()
    eps = 1e-2
    p0 = m._pflat.clone()
    m._pflat.copy_(p0 + eps * v); lp = loss_at()
    m._pflat.copy_(p0 - eps * v); lm_ = loss_at()
    m._pflat.copy_(p0)
    fd = (lp - lm_) / (2 * eps)
    an = float((g.double() @ v.double()))
    assert abs(fd - an) / (abs(an) + 1e-8) < 2e-2, (fd, an)


def test_bf16_dropout_step_runs_and_is_seeded():
    kw = dict(max_signal_length=1000, patch_size=20, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    conf = E.EcgVitConfig(**kw)  # reference default dropout 0.1 / 0.1
    m = E.EcgVit(config=conf, compute_dtype=BF16).cuda().train()
    x, y = O.synthetic_batch(8, length=1000, seed=7)
    x, y = x.cuda(), y.cuda()
    eng = m._engine()
    a = float(eng.forward(x, y, None, training=True, seed=11)[2])
    b = float(eng.forward(x, y, None, training=True, seed=11)[2])
    c = float(eng.forward(x, y, None, training=True, seed=12)[2])
    assert a == b and a != c
    ts = E.HipTrainStep(m, dict(n_step=20), sync_nonfinite=True)
    l0 = float(ts.step(x, y)[0])
    for _ in range(10):
        l1 = float(ts.step(x, y)[0])
    assert np.isfinite(l1) and l1 < l0   # overfits one batch


def test_nonfinite_gradient_raises_like_reference():
    z, m = build('g2560', F32)
    m.train()
    x, y = torch.from_numpy(z['x']).cuda(), torch.from_numpy(z['y']).cuda()
    x[0, 0, 0] = float('nan')
    ts = E.HipTrainStep(m, dict(n_step=10), sync_nonfinite=True)
    before = m._pflat.clone()
    with pytest.raises(RuntimeError, match='non-finite'):
        ts.step(x, y)
    assert torch.equal(before, m._pflat)


def test_checkpoint_roundtrip_with_oracle_state_dict():
    z, m = build('g2560', F32)
    ref = O.OracleEcgVit(config=Cfg(**load_micro('g2560')[1]))
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()}, strict=True)   # our checkpoint -> reference layout
    m2 = E.EcgVit(config=m.config)
    m2.load_state_dict(ref.state_dict(), strict=True)                                   # and back
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


def test_patch_embedding_submodule_call():
    """reference ecg_vit.py:277: `ev.vit.to_patch_embedding(x.unsqueeze(-2))`"""
    z, m = build('g2560', F32)
    x = torch.from_numpy(z['x']).cuda()
    tok = m.vit.to_patch_embedding(x.unsqueeze(-2))
    assert rel_err(tok, torch.from_numpy(z['inter/embed'])) < 1e-5


# ------------------------------------------------------------------------------------------------------ masked pre-train objective (a15)
@pytest.mark.parametrize('dtype', [F32, BF16])
def test_masked_pretrain_matches_oracle(dtype):
    """build's own SimMIM-style objective (absent from the reference => parity unpinned): HIP path vs its CPU restatement,
    same weights, same host-generated int32 mask indices (index handling bit-exact: the gathered targets must be equal)."""
    kw = dict(max_signal_length=1000, patch_size=20, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    conf = E.EcgVitConfig(**kw)
    torch.manual_seed(5)
    ref = O.OracleMaskedEcgVit(O.OracleEcgVit(config=conf)).train()
    m = E.MaskedEcgVit(E.EcgVit(config=conf, compute_dtype=dtype))
    m.load_state_dict(ref.state_dict(), strict=True)
    m.cuda().train()
    x, _ = O.synthetic_batch(6, length=1000, seed=9)
    g = torch.Generator().manual_seed(3)
    idx = m.random_mask_indices(6, generator=g)
    assert idx.shape == (6, 25) and idx.dtype == torch.int32
    o_ref = ref(x, idx)
    o_ref.loss.backward()
    out = m(x.cuda(), idx)
    out.loss.backward()
    tol = 1e-4 if dtype == F32 else 3e-2
    assert abs(float(out.loss) - float(o_ref.loss)) / float(o_ref.loss) < tol
    assert out.logits.shape == (6, 25, 240)
    assert rel_err(out.logits, o_ref.logits) < (1e-4 if dtype == F32 else 3e-2)
    eng = m.encoder._engine()
    tgt = O.patch_gather(x, 20)[torch.arange(6).unsqueeze(-1), idx.long()].reshape(-1, 240)
    assert torch.equal(eng.act['target'].float().cpu(), tgt.to(dtype).float())          # bit-exact gather of the masked patches
    pm, pr = dict(m.named_parameters()), dict(ref.named_parameters())
    for k, q in pr.items():
        p = pm[k]
        if q.grad is None:      # cls_token / classification head: untouched by this objective
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if k == 'encoder.vit.pos_embedding':
            assert float(p.grad[0, 0].abs().max()) == 0.0    # CLS position slot takes no part
        e = rel_err(p.grad, q.grad)
        assert e < (1e-4 if dtype == F32 else 0.2), (k, e)
    # fused masked train step runs and learns
    ts = E.HipTrainStep(m, dict(n_step=40, learning_rate=1e-3), sync_nonfinite=True)
    l0 = float(ts.step_masked(x.cuda(), idx)[0])
    for _ in range(15):
        l1 = float(ts.step_masked(x.cuda(), idx)[0])
    assert np.isfinite(l1) and l1 < l0
    # the supervised path still works on the shared encoder afterwards
    y = (torch.rand(6, 71) < 0.05).float().cuda()
    assert torch.isfinite(m.encoder(sample_values=x.cuda(), labels=y).loss)


# ------------------------------------------------------------------------------------------------------ data parallel on the HIP path
def _ddp_worker(rank, world, port, out_dir, overlap):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)   # both ranks share cuda:0 here; gloo moves CUDA tensors via the host
    import ecg_representation_learning_amd as E
    from oracle import vit_oracle as O
    torch.cuda.set_device(0)
    conf = E.EcgVitConfig(max_signal_length=400, patch_size=20, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(11)
    m = E.EcgVit(config=conf, compute_dtype=torch.float32).cuda().train()
    x, y = O.synthetic_batch(8, length=400, seed=77)
    lo, hi = E.ddp.shard_range(8, rank, world)
    ts = E.HipTrainStep(m, dict(n_step=10, warmup_ratio=0.0, learning_rate=1e-3), sync_nonfinite=True, overlap_allreduce=overlap)
    losses = []
    for _ in range(2):
        loss, _ = ts.step(x[lo:hi].cuda(), y[lo:hi].cuda())
        losses.append(float(loss))
    torch.save(dict(p=m._pflat.cpu(), norm=ts.grad_norm(), losses=losses), os.path.join(out_dir, f'r{rank}_{int(overlap)}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_data_parallel_step_equals_full_batch(tmp_path, overlap):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'r{r}_{int(overlap)}.pt')) for r in range(2))
    assert torch.equal(r0['p'], r1['p'])                         # replicas stay identical
    # single process, whole batch
    conf = E.EcgVitConfig(max_signal_length=400, patch_size=20, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(11)
    m = E.EcgVit(config=conf, compute_dtype=F32).cuda().train()
    x, y = O.synthetic_batch(8, length=400, seed=77)
    ts = E.HipTrainStep(m, dict(n_step=10, warmup_ratio=0.0, learning_rate=1e-3), sync_nonfinite=True)
    for _ in range(2):
        loss, _ = ts.step(x.cuda(), y.cuda())
    assert abs(ts.grad_norm() - r0['norm']) / r0['norm'] < 1e-4    # clip norm computed on the REDUCED gradient
    assert max_err(m._pflat, r0['p']) < 5e-6
    assert abs(0.5 * (r0['losses'][1] + r1['losses'][1]) - float(loss)) < 1e-5


def _ddp_masked_worker(rank, world, port, out_dir, overlap):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import ecg_representation_learning_amd as E
    from oracle import vit_oracle as O
    torch.cuda.set_device(0)
    conf = E.EcgVitConfig(max_signal_length=400, patch_size=20, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(11)
    m = E.MaskedEcgVit(E.EcgVit(config=conf, compute_dtype=torch.float32), mask_ratio=0.5).cuda().train()
    x, _ = O.synthetic_batch(8, length=400, seed=77)
    idx = m.random_mask_indices(8, generator=torch.Generator().manual_seed(5))
    lo, hi = E.ddp.shard_range(8, rank, world)
    ts = E.HipTrainStep(m, dict(n_step=10, warmup_ratio=0.0, learning_rate=1e-3), sync_nonfinite=True, overlap_allreduce=overlap)
    losses = []
    for _ in range(2):
        loss, _ = ts.step_masked(x[lo:hi].cuda(), idx[lo:hi])
        losses.append(float(loss))
    torch.save(dict(p=m.encoder._pflat.cpu(), norm=ts.grad_norm(), losses=losses), os.path.join(out_dir, f'm{rank}_{int(overlap)}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_masked_step_equals_full_batch(tmp_path, overlap):
    """the data-parallel MASKED pre-train step: two ranks on equal shards (their mean-L1 gradients summed, 1/world in the optimiser)
    = one process on the whole batch, both overlap modes"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_ddp_masked_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f'm{r}_{int(overlap)}.pt')) for r in range(2))
    assert torch.equal(r0['p'], r1['p'])
    conf = E.EcgVitConfig(max_signal_length=400, patch_size=20, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(11)
    m = E.MaskedEcgVit(E.EcgVit(config=conf, compute_dtype=F32), mask_ratio=0.5).cuda().train()
    x, _ = O.synthetic_batch(8, length=400, seed=77)
    idx = m.random_mask_indices(8, generator=torch.Generator().manual_seed(5))
    ts = E.HipTrainStep(m, dict(n_step=10, warmup_ratio=0.0, learning_rate=1e-3), sync_nonfinite=True)
    for _ in range(2):
        loss, _ = ts.step_masked(x.cuda(), idx)
    assert abs(ts.grad_norm() - r0['norm']) / r0['norm'] < 1e-4
    # two AdamW steps at lr 1e-3: an update is lr * m / (sqrt(v) + eps), so the f32 summation-order difference between "two shard sums added"
    # and "one sum over the batch" moves a weight by at most a few per cent of ONE step where the gradient is tiny (L1's sign gradients)
    assert max_err(m.encoder._pflat, r0['p']) < 2e-5
    assert abs(0.5 * (r0['losses'][1] + r1['losses'][1]) - float(loss)) < 1e-5


# ------------------------------------------------------------------------------------------------------ f2: fused input transforms
def test_fused_input_transforms_match_reference_pipeline():
    """Normalize + TimeEndPad (+ TimeOut) fused into the patch gather == the reference's host pipeline (golden transforms.npz
    holds the outputs of the reference's own Normalize / TimeEndPad incl. the full-extra-patch quirk when L % k == 0)"""
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'transforms.npz'))
    sig, mean, std = z['sig'], z['norm_mean'], z['norm_std']          # (2, 12, 50)
    xf = E.FusedInputTransform(mean, std, patch_size=25)
    assert xf.padded_length(50) == 75 and E.FusedInputTransform(mean, std, 20).padded_length(50) == 60   # quirk + plain case
    from ecg_representation_learning_amd import hip as H
    for k in (20, 25, 64):
        xf = E.FusedInputTransform(mean, std, patch_size=k)
        L = xf.padded_length(50)
        ref = np.pad((sig - mean.reshape(1, -1, 1)) / std.reshape(1, -1, 1), [(0, 0), (0, 0), (0, L - 50)])   # Normalize then pad
        assert ref.shape[-1] == z[f'pad_k{k}'].shape[-1]               # the reference's TimeEndPad output length
        np.testing.assert_array_equal(z[f'pad_k{k}'][..., :50], sig)
        np.testing.assert_allclose(ref[..., :50], z['norm_out'], rtol=1e-6)
        if k == 25:   # C*P = 300 is no multiple of 8: not a model shape, but the C-ABI kernel itself takes it (f32 patches)
            x, mu, isd = torch.from_numpy(sig).cuda(), torch.from_numpy(mean).cuda(), (1.0 / torch.from_numpy(std)).cuda()
            out = torch.empty(2 * (L // k), k * 12, device='cuda')
            assert H.lib().ecgvit_patch_gather_transform(x.data_ptr(), out.data_ptr(), 2, 12, 50, L, k, k * 12, mu.data_ptr(), isd.data_ptr(),
                                                         None, None, H.F32, H.stream()) == 0
            np.testing.assert_allclose(out.cpu().numpy().reshape(2, L // k, k * 12), O.patch_gather_np(ref.astype(np.float32), k),
                                       rtol=1e-5, atol=1e-6)
            continue
        conf = E.EcgVitConfig(max_signal_length=L, patch_size=k, hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                              intermediate_size=64, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
        m = E.EcgVit(config=conf).cuda().eval()
        m.set_input_transform(xf)
        eng = m._engine()
        with torch.no_grad():
            m(sample_values=torch.from_numpy(sig).cuda())
        got = eng.act['patches'].float().cpu().numpy().reshape(2, L // k, k * 12)
        want = O.patch_gather_np(ref.astype(np.float32), k)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    # TimeOut: seeded host draw, zeroed span, same RNG calls as the reference's TimeOut.__call__
    xf = E.FusedInputTransform(mean, std, patch_size=20, timeout=True)
    L = xf.padded_length(50)
    conf = E.EcgVitConfig(max_signal_length=L, patch_size=20, hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                          intermediate_size=64, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    m = E.EcgVit(config=conf).cuda().train()
    m.set_input_transform(xf)
    eng = m._engine()
    torch.manual_seed(5)
    m(sample_values=torch.from_numpy(sig).cuda())
    got = eng.act['patches'].float().cpu().numpy().reshape(2, L // 20, 240)
    torch.manual_seed(5)
    ref = np.pad((sig - mean.reshape(1, -1, 1)) / std.reshape(1, -1, 1), [(0, 0), (0, 0), (0, L - 50)]).astype(np.float32)
    sampler = torch.distributions.Uniform(low=0.0, high=0.5)
    for b in range(2):
        r = sampler.sample().item()
        l_crop = round(r * L)
        st = torch.randint(high=L - l_crop, size=(1,)).item()
        ref[b, :, st:st + l_crop] = 0
    np.testing.assert_allclose(got, O.patch_gather_np(ref, 20), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------------ f1: evaluate on device
def test_evaluator_matches_oracle_eval_pass():
    """HipEvaluator (MyTrainer.evaluate on the device) vs the oracle model + oracle metrics on the same weights and records"""
    from oracle.metrics_oracle import get_accuracy_np
    kw = dict(max_signal_length=2560, patch_size=64, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128)
    conf, om, m, x, y = _oracle_pair(kw, 24, F32)
    ev = E.HipEvaluator(m, eval_batch_size=10)                          # ragged last batch (10, 10, 4)
    m.train()
    out = ev.evaluate(x.cuda(), y.cuda(), return_predictions=True)
    assert m.training                                                   # mode restored, as the reference does
    om.eval()
    with torch.no_grad():
        outs = [om(sample_values=x[s:s + 10], labels=y[s:s + 10]) for s in range(0, 24, 10)]
    lo, loss = torch.cat([o.logits for o in outs]), float(np.mean([float(o.loss) for o in outs]))
    assert max_err(out['predictions']['logits'], lo) < 1e-4
    want = get_accuracy_np(torch.sigmoid(out['predictions']['logits']).cpu().numpy(), y.numpy())
    d = out['metrics']
    assert abs(d['eval/loss'] - loss) / loss < 1e-5
    for k in ('binary_accuracy', 'weighted_binary_accuracy', 'binary_negative_recall', 'binary_positive_recall'):
        assert abs(d[f'eval/{k}'] - want[k]) < 1e-12
    assert (d['eval/macro_auc'] is None) == (want['macro_auc'] is None)
    if want['macro_auc'] is not None:
        assert abs(d['eval/macro_auc'] - want['macro_auc']) < 1e-6      # in-kernel f32 sigmoid vs torch's: tie structure only


# ------------------------------------------------------------------------------------------------------ f3: checkpoints, attention export
def test_load_trained_reference_checkpoint_and_attention_export(tmp_path):
    """a `.pt` written the way the reference's trainer writes it (torch.save(state_dict) of the oracle model = reference key layout)
    loads strictly; the fused bf16 path's exported attention == the f32 path's materialised probabilities (bf16 q/k rounding:
    5e-3 absolute on probabilities), rows sum to 1, and the visualiser's rollout matches the same arithmetic on the oracle's probabilities"""
    torch.manual_seed(3)
    conf = E.EcgVitConfig.from_defined('ecg-vit-tiny')
    ref = O.OracleEcgVit(config=conf)
    path = os.path.join(tmp_path, 'model.pt')
    torch.save(ref.state_dict(), path)
    bad = {k: v for k, v in ref.state_dict().items() if 'mlp_head' not in k}
    torch.save(bad, os.path.join(tmp_path, 'bad.pt'))
    with pytest.raises(RuntimeError):
        E.load_trained('ecg-vit-tiny', os.path.join(tmp_path, 'bad.pt'))           # strict, like the reference
    m32 = E.load_trained('ecg-vit-tiny', path).cuda()
    assert not m32.training
    x, _ = O.synthetic_batch(2, length=conf.max_signal_length, seed=5)
    with torch.no_grad():
        o32 = m32(sample_values=x.cuda())
    ref.eval()
    with torch.no_grad():
        assert max_err(o32.logits, ref(sample_values=x).logits) < 1e-4
    if True:
        m16 = E.load_trained('ecg-vit-tiny', path, compute_dtype=BF16).cuda()
        with torch.no_grad():
            m16(sample_values=x.cuda())
        for layer in range(conf.num_hidden_layers):
            p16, p32 = m16.attention_probs(layer), m32.attention_probs(layer)
            assert p16.shape == p32.shape
            assert float((p16.sum(-1) - 1).abs().max()) < 1e-3
            assert float((p16 - p32).abs().max()) < 5e-3
    logits, amap = m32.attention_rollout(x[0].cuda())
    L = conf.num_hidden_layers
    attn = torch.stack([m32.attention_probs(i)[0].mean(0) for i in range(L)]).cpu()   # probs of the single-record forward it ran
    attn = attn + torch.eye(attn.size(1))
    attn = attn / attn.sum(-1, keepdim=True)
    res = torch.stack([attn[0]] + [attn[i] @ attn[i - 1] for i in range(1, L)])[:, 0, 1:]
    res = res / res.max()
    assert amap.shape == (L, conf.max_signal_length // conf.patch_size)
    assert max_err(amap, res) < 1e-5 and float(amap.max()) == 1.0
    assert max_err(logits, o32.logits[0]) < 1e-5


def test_bf16_input_gradients_via_transposed_shadows_match_plain_path():
    """dgrad on the forward kernel against W^T shadows vs the A.B path on W: same bf16 products, f32 accumulation in another order"""
    kw = dict(max_signal_length=5000, patch_size=20, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024)
    conf, ref, m, x, y = _oracle_pair(kw, 16, BF16)                      # M = 16 * 251 = 4016 rows >= 2048: the large-shape kernels
    m.train()
    eng = m._engine()
    eng.aux8 = False   # (the e4m3 saved FFN tensor needs the transposed shadows: with it the two passes below would differ by its rounding, not by the dgrad route)
    assert len(eng.WT) == 8 and eng.WT['vit.transformer.layers.0.1.fn.net.0.weight'].shape == (256, 1024)
    torch.testing.assert_close(eng.WT['vit.transformer.layers.1.0.fn.to_qkv.weight'].float().t(),
                               eng.W['vit.transformer.layers.1.0.fn.to_qkv.weight'].float(), rtol=0, atol=0)
    m(sample_values=x.cuda(), labels=y.cuda()).loss.backward()
    g_t = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    saved, eng.WT = eng.WT, {}
    m(sample_values=x.cuda(), labels=y.cuda()).loss.backward()
    eng.WT = saved
    for k, p in m.named_parameters():
        assert rel_err(g_t[k], p.grad) < 3e-3, (k, rel_err(g_t[k], p.grad))
    # and after a fused optimiser step the transposed shadows follow the updated weights
    ts = E.HipTrainStep(m, dict(n_step=20), sync_nonfinite=True)
    ts.step(x.cuda(), y.cuda())
    ts.finish()
    for k, wt in eng.WT.items():
        assert torch.equal(wt.t(), eng.W[k]) and torch.equal(eng.W[k], eng.P32[k].to(BF16)), k


def test_activation_pool_survives_batch_size_switches():
    """train (B = 8) / eval (B = 3) alternation and a short last batch: the activation slabs come out of ONE pool sized for the largest batch seen -- a
    smaller batch takes prefix views of it (same storage, nothing re-requested from the allocator), results equal a fresh engine's, and only a LARGER
    batch (or another objective) re-makes the pool"""
    kw = dict(max_signal_length=1000, patch_size=20, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    conf, ref, m, x, y = _oracle_pair(kw, 8, BF16)
    m.train()
    eng = m._engine()
    xc, yc = x.cuda(), y.cuda()
    l8 = float(m(sample_values=xc, labels=yc).loss.detach())
    base = {k: v.data_ptr() for k, v in eng._pool.items()}
    p_qkv = eng.act['layers'][1]['qkv'].data_ptr()
    m.eval()
    with torch.no_grad():
        lg3 = m(sample_values=xc[:3].contiguous()).logits.clone()
    assert eng.B == 3 and eng.act['layers'][1]['qkv'].data_ptr() == p_qkv and eng.act['layers'][1]['qkv'].shape[0] == 3 * eng.N
    assert {k: v.data_ptr() for k, v in eng._pool.items()} == base                  # nothing was re-allocated
    m.train()
    out = m(sample_values=xc, labels=yc)
    out.loss.backward()
    assert float(out.loss.detach()) == l8 and {k: v.data_ptr() for k, v in eng._pool.items()} == base
    m2 = E.EcgVit(config=conf, compute_dtype=BF16)
    m2.load_state_dict(ref.state_dict())
    m2.cuda().eval()
    with torch.no_grad():
        assert torch.equal(m2(sample_values=xc[:3].contiguous()).logits, lg3)      # a fresh engine, sized for 3 records, computes the same bits
    x12, y12 = O.synthetic_batch(12, length=1000, seed=5)
    assert torch.isfinite(m(sample_values=x12.cuda(), labels=y12.cuda()).loss)
    assert eng._pool_B == 12 and eng._pool['L1.qkv'].numel() >= 12 * eng.N * 3 * 128   # grown once, for the larger batch


def test_activation_pool_survives_crossing_the_e4m3_saved_tensor_boundary():
    """a batch sequence that crosses 2048 token rows: above, the saved FFN tensor `hpre` is e4m3 BYTES (ECGVIT_EPI_AUX8), below, bf16 -- the slab is pooled
    as bytes and viewed per pass, so the long / short / long alternation (an epoch's short last batch) re-requests NOTHING, and every pass computes what
    a fresh engine of its own size computes"""
    kw = dict(max_signal_length=1000, patch_size=20, hidden_size=192, num_hidden_layers=2, num_attention_heads=3, intermediate_size=384)
    conf, ref, m, x, y = _oracle_pair(kw, 45, BF16)       # 45 x 51 = 2295 token rows
    m.train()
    eng = m._engine()
    xc, yc = x.cuda(), y.cuda()
    big = m(sample_values=xc, labels=yc)
    big.loss.backward()
    assert eng._aux8(45 * eng.N) and eng.act['layers'][0]['hpre'].dtype == torch.uint8
    l45, g45 = float(big.loss.detach()), m._gflat.clone()
    base = {k: v.data_ptr() for k, v in eng._pool.items()}
    m.zero_grad(set_to_none=True)                                                  # (loss.backward() ACCUMULATES into .grad like any module)
    small = m(sample_values=xc[:8].contiguous(), labels=yc[:8].contiguous())      # 408 rows: the bf16 saved tensor, on the small kernels
    small.loss.backward()
    assert not eng._aux8(8 * eng.N) and eng.act['layers'][0]['hpre'].dtype == BF16 and eng.act['layers'][0]['hpre'].shape == (8 * eng.N, 384)
    assert {k: v.data_ptr() for k, v in eng._pool.items()} == base
    l8, g8 = float(small.loss.detach()), m._gflat.clone()
    m.zero_grad(set_to_none=True)
    again = m(sample_values=xc, labels=yc)
    again.loss.backward()
    assert {k: v.data_ptr() for k, v in eng._pool.items()} == base and eng.act['layers'][0]['hpre'].dtype == torch.uint8
    assert float(again.loss.detach()) == l45 and torch.equal(m._gflat, g45)
    m2 = E.EcgVit(config=conf, compute_dtype=BF16)
    m2.load_state_dict(ref.state_dict())
    m2.cuda().train()
    o2 = m2(sample_values=xc[:8].contiguous(), labels=yc[:8].contiguous())
    o2.loss.backward()
    assert float(o2.loss.detach()) == l8 and torch.equal(m2._gflat, g8)
    # the constructor option (saved_ffn_e4m3=False): the same model keeps the tensor in bf16 at every size
    m3 = E.EcgVit(config=conf, compute_dtype=BF16, saved_ffn_e4m3=False)
    m3.load_state_dict(ref.state_dict())
    m3.cuda().train()
    o3 = m3(sample_values=xc, labels=yc)
    assert m3._engine().act['layers'][0]['hpre'].dtype == BF16 and float(o3.loss.detach()) == l45      # forward values do not depend on it
    with pytest.raises(ValueError):
        E.EcgVit(config=conf, compute_dtype=F32, saved_ffn_e4m3=True)._engine()


# ------------------------------------------------------------------------------------------------------ f4: record feeding
def test_device_feeder_pinned_async_batches_and_fused_transform(tmp_path):
    """double-buffered pinned H2D feeder: every batch arrives intact and in order over two epochs while the consumer keeps the
    device busy; feeding RAW records into a model with the fused input transform == transforming on the host first"""
    rng = np.random.default_rng(1)
    n, L_raw, k = 150, 990, 20
    rec = rng.standard_normal((n, 12, L_raw)) * 3 + 1                         # float64 on disk
    path = os.path.join(tmp_path, 'rec.npy')
    np.save(path, rec)
    mh = (rng.random((n, 71)) < 0.05).astype(np.float32)
    idx = np.arange(n)[::-1].copy()                                           # non-monotonic source rows
    feeder = E.DeviceFeeder(path, idx, mh, batch_size=32, shuffle=False)
    want_x, want_y = torch.from_numpy(rec[idx].astype(np.float32)), torch.from_numpy(mh)
    for _ in range(2):
        xs, ys = [], []
        for b in feeder:
            assert b['sample_values'].is_cuda and b['sample_values'].dtype == F32
            junk = torch.randn(2048, 2048, device='cuda') @ torch.randn(2048, 2048, device='cuda')   # consumer work between batches
            xs.append(b['sample_values'].clone()); ys.append(b['labels'].clone())
        assert torch.equal(torch.cat(xs).cpu(), want_x) and torch.equal(torch.cat(ys).cpu(), want_y)
    mean, std = rec.mean(axis=(0, 2)).astype(np.float32), rec.std(axis=(0, 2)).astype(np.float32)
    xf = E.FusedInputTransform(mean, std, patch_size=k)
    L = xf.padded_length(L_raw)
    conf = E.EcgVitConfig(max_signal_length=L, patch_size=k, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, hidden_dropout_prob=0., attention_probs_dropout_prob=0.)
    torch.manual_seed(0)
    m_raw = E.EcgVit(config=conf).cuda().eval()
    m_ref = E.EcgVit(config=conf).cuda().eval()
    m_ref.load_state_dict(m_raw.state_dict())
    m_raw.set_input_transform(xf)
    b = next(iter(feeder))
    host = np.pad((rec[idx[:32]] - mean.reshape(1, -1, 1)) / std.reshape(1, -1, 1), [(0, 0), (0, 0), (0, L - L_raw)]).astype(np.float32)
    with torch.no_grad():
        o_raw = m_raw(sample_values=b['sample_values'], labels=b['labels'])
        o_ref = m_ref(sample_values=torch.from_numpy(host).cuda(), labels=b['labels'])
    assert max_err(o_raw.logits, o_ref.logits.cpu()) < 1e-4 and abs(float(o_raw.loss) - float(o_ref.loss)) < 1e-5


def test_training_overfits_one_batch_on_the_large_shape_kernels():
    """end-to-end: 40 fused steps on ONE fixed batch at a geometry that runs every large-shape kernel (persistent A.B^T GEMM for forward
    and input gradients, streaming weight-gradient GEMM, persistent attention backward with N = 251 > 128, exact-fit LayerNorm,
    dropout 0.1): the bf16 path must drive the loss down like the f32 parity path does"""
    kw = dict(max_signal_length=5000, patch_size=20, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
              hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    x, y = O.synthetic_batch(16, length=5000, seed=3)
    curves = {}
    for dtype in (BF16, F32):
        torch.manual_seed(11)
        m = E.EcgVit(config=E.EcgVitConfig(**kw), compute_dtype=dtype).cuda().train()
        ts = E.HipTrainStep(m, dict(n_step=40, learning_rate=1e-3, schedule='constant'), sync_nonfinite=True)
        torch.manual_seed(5)                                   # same dropout seeds for both paths
        losses = [float(ts.step(x.cuda(), y.cuda())[0]) for _ in range(40)]
        ts.finish()
        assert all(np.isfinite(losses))
        curves[dtype] = losses
    b, f = curves[BF16], curves[F32]
    assert f[-1] < 0.5 * f[0] and b[-1] < 0.5 * b[0], (b[0], b[-1], f[0], f[-1])          # it learns
    assert abs(b[0] - f[0]) / f[0] < 2e-2
    assert abs(np.mean(b[-5:]) - np.mean(f[-5:])) < 0.05 * max(1.0, f[0]), (b[-5:], f[-5:])   # same trajectory within bf16 / mask noise


def test_bf16_long_record_geometry_trains_seq_500():
    """patch 10 on 5000 samples: N = 501 tokens.  The fused bf16 forward covers N <= 512; the backward beyond 256 tokens recomputes the
    probabilities with the exact-f32 batched kernels.  Gradients vs the f32 parity path (cosine >= 0.98, as for every bf16 check), and a
    fused step with dropout runs, is finite and reproducible for a fixed seed"""
    kw = dict(max_signal_length=5000, patch_size=10, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    conf, ref, m16, x, y = _oracle_pair(kw, 4, BF16)
    m32 = E.EcgVit(config=conf, compute_dtype=F32).cuda()
    m32.load_state_dict(m16.state_dict())
    m16.train(); m32.train()
    o16 = m16(sample_values=x.cuda(), labels=y.cuda()); o16.loss.backward()
    o32 = m32(sample_values=x.cuda(), labels=y.cuda()); o32.loss.backward()
    assert abs(float(o16.loss) - float(o32.loss)) / float(o32.loss) < 2e-2
    for (k, p), (_, q) in zip(m16.named_parameters(), m32.named_parameters()):
        cos = torch.nn.functional.cosine_similarity(p.grad.flatten().double(), q.grad.flatten().double(), dim=0)
        assert float(cos) > 0.98, (k, float(cos))
    conf_d = E.EcgVitConfig(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **kw)
    runs = []
    for _ in range(2):
        torch.manual_seed(3)
        md = E.EcgVit(config=conf_d, compute_dtype=BF16).cuda().train()
        ts = E.HipTrainStep(md, dict(n_step=10), sync_nonfinite=True)
        torch.manual_seed(4)
        ls = [float(ts.step(x.cuda(), y.cuda())[0]) for _ in range(3)]
        ts.finish()
        runs.append((ls, md._pflat.clone()))
    assert all(np.isfinite(runs[0][0])) and runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
