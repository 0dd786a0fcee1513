"""-m gpu: the data-parallel train step's RCCL path on ONE GPU.

A 1-rank `nccl` (= RCCL) process group with `single_rank_collectives=True` makes HipTrainStep issue everything the N > 1 path issues --
the start broadcast of the flat parameter buffer, one asynchronous all-reduce per gradient bucket from inside the backward pass (on
RCCL's stream, overlapped with the remaining backward kernels, which are launched as dispatcher-balanced chunks meanwhile), the bf16
staging copies, the waits in front of the fused norm + clip + AdamW -- with a sum over one rank as the arithmetic.  The result must
therefore equal the plain single-process step: bit for bit with f32 on the wire, within bf16 rounding with bf16 on the wire.
(Multi-rank SEMANTICS -- shards, sum, 1/world -- are pinned by the world_size-2 gloo test on CPU, tests/test_ddp_gloo.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import ecg_representation_learning_amd as E

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope='module')
def nccl_group():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield None
    dist.destroy_process_group()


def _model(seed=5):
    conf = E.EcgVitConfig(max_signal_length=5000, patch_size=20, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                          hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(seed)
    return E.EcgVit(config=conf, compute_dtype=BF16).cuda().train()


def _run(steps, **kw):
    m = _model()
    x, y = E.workload.synthetic_batch(24, length=5000, seed=3)
    x, y = x.cuda(), y.cuda()
    st = E.HipTrainStep(m, dict(n_step=20, warmup_ratio=0.0), **kw)
    torch.manual_seed(99)               # the host RNG draws the per-step dropout seeds
    losses = [float(st.step(x, y)[0]) for _ in range(steps)]
    st.finish()
    torch.cuda.synchronize()
    return losses, m._pflat.clone(), st.grad_norm()


def test_rccl_path_on_one_rank_equals_plain_step(nccl_group):
    ref_l, ref_p, ref_n = _run(3)                                              # collectives skipped (1 rank)
    # f32 on the wire: a sum over one rank is the identity, and the chunked GEMM launches of the overlapped backward hand the same
    # tiles to other workgroups without changing any sum -- bit-identical in both modes
    for kw in (dict(overlap_allreduce=False), dict(overlap_allreduce=True)):
        l, p, n = _run(3, single_rank_collectives=True, **kw)
        assert l == ref_l and torch.equal(p, ref_p) and n == ref_n, kw
    for kw in (dict(overlap_allreduce=True), dict(overlap_allreduce=False)):
        l, p, n = _run(3, single_rank_collectives=True, grad_comm_dtype=BF16, **kw)
        # tolerance: the gradient passes through bf16 once (2^-9 relative per element) before AdamW; three steps at lr 3e-4
        assert all(abs(a - b) < 2e-3 * abs(b) + 1e-4 for a, b in zip(l, ref_l)), (l, ref_l)
        assert float((p - ref_p).norm() / ref_p.norm()) < 1e-3
        assert abs(n - ref_n) < 1e-2 * ref_n


def _run_masked(steps, **kw):
    m = E.MaskedEcgVit(_model(), mask_ratio=0.5).cuda().train()
    x, _ = E.workload.synthetic_batch(24, length=5000, seed=3)
    x = x.cuda()
    idx = m.random_mask_indices(24, generator=torch.Generator().manual_seed(8))
    st = E.HipTrainStep(m, dict(n_step=20, warmup_ratio=0.0), **kw)
    torch.manual_seed(99)
    losses = [float(st.step_masked(x, idx)[0]) for _ in range(steps)]
    st.finish()
    torch.cuda.synchronize()
    return losses, m.encoder._pflat.clone(), st.grad_norm()


def test_rccl_masked_step_on_one_rank_equals_plain_step(nccl_group):
    """the MASKED pre-train step (the north_star's data-parallel loop is the pre-train step) under a process group: head bucket released at
    the top of the backward pass, layer buckets from inside the trunk, embed + pretrain at its end -- through RCCL on one rank, both overlap
    modes, bit-identical to the plain masked step with f32 on the wire"""
    ref_l, ref_p, ref_n = _run_masked(3)
    assert ref_l[2] < ref_l[0]
    for kw in (dict(overlap_allreduce=False), dict(overlap_allreduce=True)):
        l, p, n = _run_masked(3, single_rank_collectives=True, **kw)
        assert l == ref_l and torch.equal(p, ref_p) and n == ref_n, kw
    l, p, n = _run_masked(3, single_rank_collectives=True, grad_comm_dtype=BF16, overlap_allreduce=True)
    assert all(abs(a - b) < 2e-3 * abs(b) + 1e-4 for a, b in zip(l, ref_l)), (l, ref_l)
    assert float((p - ref_p).norm() / ref_p.norm()) < 1e-3


def test_masked_backward_releases_buckets_in_ready_order(nccl_group):
    """every bucket of the layout is reported exactly once per masked backward, `head` (zero for this objective) first"""
    m = E.MaskedEcgVit(_model(), mask_ratio=0.5).cuda().train()
    x, _ = E.workload.synthetic_batch(8, length=5000, seed=3)
    idx = m.random_mask_indices(8, generator=torch.Generator().manual_seed(8))
    eng = m.encoder._engine()
    eng.forward_masked(x.cuda(), idx.cuda(), training=True, seed=1)
    seen = []
    eng.on_grads_ready = seen.append
    try:
        eng.backward_masked()
    finally:
        eng.on_grads_ready = None
    want = [k for k, _ in m.encoder._layout.buckets_in_ready_order(eng.Ly)]
    assert seen == want and seen[0] == 'head' and seen[-1] == 'pretrain', (seen, want)


def test_grad_exchange_streams_on_gpu(nccl_group):
    """GradExchange alone on device buffers: buckets reduced on RCCL's stream while the producer stream keeps writing later buckets"""
    n = 1 << 22
    ranges = [(f'b{i}', (i * (n // 8), (i + 1) * (n // 8))) for i in range(8)]
    for kw in (dict(comm_dtype=torch.float32), dict(comm_dtype=BF16)):
        g = torch.zeros(n, device='cuda')
        ex = E.ddp.GradExchange(ranges, overlap=True, single_rank_collectives=True, **kw)
        ex.begin(g)
        for i, (tag, (lo, hi)) in enumerate(ranges):
            g[lo:hi] = float(i + 1) * 0.5          # values exactly representable in bf16
            ex.bucket_ready(tag)
        ex.finish()
        torch.cuda.synchronize()
        want = torch.cat([torch.full((n // 8,), float(i + 1) * 0.5) for i in range(8)]).cuda()
        assert torch.equal(g, want), kw
    with pytest.raises(RuntimeError):
        ex = E.ddp.GradExchange(ranges, overlap=True, single_rank_collectives=True)
        ex.begin(torch.zeros(n, device='cuda'))
        ex.bucket_ready('b0')
        ex.finish()                                # seven buckets never reported


def test_failed_overlapped_step_leaves_no_launch_geometry_behind(nccl_group, monkeypatch):
    """`tiles_per_workgroup` is an argument of ONE backward pass (engine.backward(..., tiles_per_workgroup=)), not process state: a
    step that dies inside the overlapped backward -- here the gradient exchange never hears of a bucket and `finish()` raises, after an
    exception thrown from inside the backward itself -- must leave the next plain step (and any other model in the process) on
    persistent launches."""
    from ecg_representation_learning_amd import hip
    seen = []
    real = hip.gemm

    def spy(layout, *a, **k):
        seen.append(k.get('tiles_per_workgroup', 0))
        return real(layout, *a, **k)
    monkeypatch.setattr(hip, 'gemm', spy)
    assert not hasattr(hip, 'GEMM_TILES_PER_WORKGROUP')          # the process-wide knob is gone
    m = _model()
    x, y = E.workload.synthetic_batch(24, length=5000, seed=3)
    x, y = x.cuda(), y.cuda()
    st = E.HipTrainStep(m, dict(n_step=20, warmup_ratio=0.0), single_rank_collectives=True, overlap_allreduce=True)
    st.step(x, y)
    assert 2 in seen and m._engine()._tpw == 0                   # the overlapped backward ran chunked; the setting ended with the pass
    eng = m._engine()
    boom = RuntimeError('injected failure inside the backward pass')
    real_ready = eng._ready

    def failing_ready(tag):
        if tag == 'layer0':
            raise boom
        real_ready(tag)
    eng._ready = failing_ready
    with pytest.raises(RuntimeError, match='injected failure'):
        st.step(x, y)
    eng._ready = real_ready
    torch.cuda.synchronize()
    assert eng._tpw == 0 and eng.on_grads_ready is None
    # the next PLAIN step of the same process: persistent launches only
    seen.clear()
    plain = E.HipTrainStep(_model(seed=6), dict(n_step=20, warmup_ratio=0.0))
    plain.step(x, y)
    plain.finish()
    torch.cuda.synchronize()
    assert seen and set(seen) == {0}


def test_bench_self_launcher_single_rank_collectives():
    """`python bench.py --gpus 1 --single-rank-collectives` goes through the SAME launcher code `--gpus 8` would use without torchrun:
    the parent starts the rank processes before touching the GPU, relays rank 0's JSON line, and reports the RCCL group size."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--single-rank-collectives', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-masked', '--no-bf16-saved', '--no-small', '--no-fp8-large', '--batch', '64'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['rccl_ranks'] == 1 and out['n_gpus'] == 1 and len(out['rank_ms_per_step']) == 1
    assert abs(max(out['rank_ms_per_step']) - out['ms_per_step']) <= 1e-9 * out['ms_per_step']   # the headline run's own per-rank time
    assert out['config']['parallelism'] == 'dp1+single-rank-collectives' and out['value'] > 0
