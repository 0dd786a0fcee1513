#!/usr/bin/env python3
"""
bench.py -- 12-lead ECG records/sec for one train step of EcgVit (BASELINE.json metric), on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W                         (single process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher: the parent starts N rank processes itself, before touching the GPU, relays
                                       rank 0's JSON line and exits non-zero if any rank failed)

A "step" = the reference's step body (ecg_transformer/models/train.py:271-283) on one synthetic batch already
resident in HBM: forward (patch-embed -> L x [LN, MHSA, LN, GELU-FFN] -> head) + BCE loss + backward + global-norm
clip + AdamW [+ RCCL gradient all-reduce for N > 1], dropout active at the reference's default 0.1.
Workload at every N: EcgVit-base, bf16 MFMA path, 512 records x 12 leads x 5000 samples per GPU, patch 20 (251 tokens)
(BASELINE.json configs[2]/[3]; weak scaling: global batch = 512 * N).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (gemm_nt_kernel: the bf16 MFMA A.B^T GEMM of the Linear forward and input-gradient products):
                algorithmic FLOPs of its launches / their HIP-event time measured inside the timed region, against the
                2.5 PFLOP/s dense bf16 peak; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes, null
                (with the reason in `traffic_source`) when the kernel sources changed since that profile was taken.
  bf16_saved_tensor  (N = 1, default workload) the SAME step with the saved FFN tensor kept in bf16 (saved_ffn_e4m3=False) instead of e4m3 bytes: the headline
                step stores one backward-only tensor per layer narrower than the metric's dtype; this is the pure-bf16-storage figure beside it.
  masked        (N = 1, default workload) the build's masked pre-train step timed in the same process: value, ms_per_step, model TFLOP/s at its own
                250-token FLOP count, roofline.
  small         (N = 1, default workload) BASELINE.json configs[1]: EcgVit-small bf16, 251 tokens, 256 records: value, ms_per_step, roofline.
  fp8_large     (N = 1, default workload) BASELINE.json configs[4] on one GPU: EcgVit-large / 501 tokens with fp8 Linear operands, and the
                same step with bf16 operands back to back on the same device (value, ms_per_step, fp8_over_bf16, roofline vs 5 PFLOP/s).
  cpu_baseline  the CPU oracle's train step (torch eager f32, all host cores) on a bounded sample of the same workload.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA
PEAK_F32_TFLOPS = 157.3
FP8_DESC = ('fp8 Linear operands on the block-scaled fp8 MFMA: e4m3 activations / weights in the forward products, e5m2 gradients x e4m3 weights in the input-gradient products, e5m2 gradients x e4m3 activations in the weight-gradient products (f32 split-K accumulation); attention, LayerNorm, optimiser bf16 / f32')

CONFIGS = {
    # name: (from_defined name | dict of fields, per-GPU batch)
    'base': ('ecg-vit-base', 512),
    'small': ('ecg-vit-small', 256),
    'tiny': ('ecg-vit-tiny', 256),
    'large': ('ecg-vit-large', 256),
    'tiny2': (dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512), 32),
}


def dropout_desc(conf):
    """dropout as APPLIED by the bf16 path: every site draws 8 bits per element (one hash per four consecutive elements; the fused attention
    kernels: per four keys), so p runs at round(256 p)/256 (0.1 -> 26/256 = 0.1016), kept values rescaled by the exact keep rate; the f32
    parity path applies the exact p"""
    p, pa = conf.hidden_dropout_prob, conf.attention_probs_dropout_prob

    def q(v):
        t = min(255, int(v * 256.0 + 0.5)) if v > 0 else 0
        return f'{t}/256 = {t / 256:.4f}'
    return (f'hidden dropout applied at {q(p)} (configured {p:g}), embedding dropout at {q(pa)}, attention-probability dropout at {q(p)} '
            f'(bf16 path: 8 random bits per element)')


def make_config(E, name, patch, length, dropout):
    spec, batch = CONFIGS[name]
    conf = E.EcgVitConfig.from_defined(spec) if isinstance(spec, str) else E.EcgVitConfig(**spec)
    conf.max_signal_length, conf.patch_size = length, patch
    if dropout is not None:
        conf.hidden_dropout_prob = conf.attention_probs_dropout_prob = dropout
    return conf, batch


BYTES_MODEL = 2   # what `roofline.alg_bytes_per_launch` counts. 1 (rounds 1-4): both operands + the output. 2 (round 5 on): + every stream of the
                  # epilogue (residual rows, the saved GELU' x mask tensor when one is passed, the 8-bit copy of the output).  Lines with different
                  # models are not comparable on alg_bytes / traffic ratios (tools/check_profiles.py refuses to)


class GemmProbe:
    """HIP-event timing of every launch of ONE kernel symbol (layout, out dtype) inside the timed region."""

    def __init__(self, hip, layout, out_dtype):
        self.hip, self.layout, self.out_dtype = hip, layout, out_dtype
        self.orig = hip.gemm
        self.events, self.flops, self.bytes, self.n8 = [], 0.0, 0.0, 0
        self.enabled = False

    def install(self):
        probe = self

        def gemm(layout, A, B, C, M, N, K, *a, **k):
            # which kernel symbol this call runs on is the LIBRARY's answer (ecgvit_gemm_kernel: its own dispatch, nothing launched)
            # (C is None: an 8-bit product that emits only the 8-bit copy of its bf16-rounded output -- EPI_NO_OUT)
            hit = probe.enabled and layout == probe.layout and (torch.bfloat16 if C is None else C.dtype) == probe.out_dtype and \
                probe.hip.gemm_kernel(layout, A, B, C, M, N, K, *a, **k) == probe.hip.KERNEL_GEMM_NT
            f8 = k.get('fp8_format') is not None
            if hit:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            probe.orig(layout, A, B, C, M, N, K, *a, **k)
            if hit:
                e1.record()
                probe.events.append((e0, e1))
                probe.flops += 2.0 * M * N * K
                # algorithmic bytes = every tensor the launch must read or write ONCE: both operands, the output when it is written, and what
                # the epilogue streams -- residual rows, the saved GELU' x mask tensor (written by the FFN-up forward, read by the FFN-down
                # input gradient), the 8-bit copy of the output (round 5; before, only operands + C were counted: 0.793 GB per launch at base
                # where this count gives 0.86)
                epi = k.get('epilogue', 0)
                nb = (1.0 if f8 else 2.0) * (M * K + N * K) + (0.0 if C is None else 2.0) * M * N
                nb += 2.0 * M * N * bool(epi & probe.hip.EPI_RESIDUAL)
                if k.get('aux') is not None:   # (the saved tensor streams only when the caller hands one over)
                    nb += (1.0 if epi & probe.hip.EPI_AUX8 else 2.0) * M * N * bool(epi & (probe.hip.EPI_GELU_GRAD_AUX | probe.hip.EPI_MUL_AUX | probe.hip.EPI_GELU_BWD | probe.hip.EPI_GELU))
                nb += 1.0 * M * N * bool(epi & probe.hip.EPI_QUANT_OUT)
                probe.bytes += nb
                probe.n8 += 1 if f8 else 0
        self.hip.gemm = gemm
        import ecg_representation_learning_amd.engine as eng
        self._eng_hip = eng.hip
        eng.hip.gemm = gemm

    def uninstall(self):
        self.hip.gemm = self.orig
        self._eng_hip.gemm = self.orig

    def result(self):
        if not self.events:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        n = len(self.events)
        return dict(launches=n, avg_us=1e3 * ms / n, tflops=self.flops / (ms * 1e-3) / 1e12, alg_bytes_per_launch=self.bytes / n,
                    flops_per_launch=self.flops / n, launches_8bit=self.n8)


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel sources and the C-ABI header: identifies the build a profile was taken on"""
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(src, '*.hip')) + glob.glob(os.path.join(src, '*.h')) + [os.path.join(ROOT, 'include', 'ecgvit_hip.h')]):
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def workload_key(args):
    """identifies the workload a counter profile was taken on (profiles/*.json carry it next to the kernel-source hash)"""
    _, batch = CONFIGS[args.config]
    drop = 'default' if args.dropout is None else f'{args.dropout:g}'
    return f'{args.config}-{args.dtype}-b{args.batch or batch}-p{args.patch}-l{args.length}-drop{drop}-{args.objective}'


def pmc_traffic(kernel_key, args):
    """(HBM bytes per launch of the dominant kernel, where that number comes from).  Counters cannot be read from inside the
    process: the value is the one a committed rocprofv3 --pmc profile measured on this workload (tools/pmc_bench.sh:
    2 x FETCH_SIZE per the gfx950 calibration + WRITE_SIZE) -- reported only while the kernel sources are the ones profiled."""
    key, sha = workload_key(args), kernel_source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_*.json')), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
        except Exception:
            continue
        if prof.get('workload_key') != key:
            continue
        name = os.path.basename(path)
        if prof.get('kernel_source_sha16') != sha:
            stale = stale or f'none: kernel sources changed since profiles/{name} (taken on sources {prof.get("kernel_source_sha16")})'
            continue
        try:
            return prof['kernels'][kernel_key]['hbm_bytes_per_launch'], f'profiles/{name} @ sources {sha}'
        except KeyError:
            return None, f'none: profiles/{name} has no entry for {kernel_key}'
    return None, stale or f'none: no counter profile of workload {key} under profiles/'


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(conf, seconds_budget=25.0, masked=False):
    """the CPU oracle (torch eager f32 restatement of the reference step) on the host cores, bounded sample (~seconds_budget)"""
    from oracle import vit_oracle as O   # the ONLY place bench.py touches the oracle: the timed CPU baseline
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    flops_rec = O.train_flops_per_record(conf)
    torch.manual_seed(77)
    model = O.OracleEcgVit(config=conf).train()
    n_patch = conf.max_signal_length // conf.patch_size
    if masked:
        mm = O.OracleMaskedEcgVit(model).train()

        class _Adapter(torch.nn.Module):   # OracleTrainer calls model(sample_values=, labels=): labels carry the mask indices
            def __init__(self):
                super().__init__()
                self.mm = mm

            def forward(self, sample_values, labels):
                return self.mm(sample_values, labels)
        tr = O.OracleTrainer(_Adapter(), n_step=100)
    else:
        tr = O.OracleTrainer(model, n_step=100)

    def batch_of(b):
        xx, yy = O.synthetic_batch(b, length=conf.max_signal_length, seed=77)
        if masked:
            yy = torch.stack([torch.randperm(n_patch)[:n_patch // 2] for _ in range(b)]).int()
        return xx, yy
    x1, y1 = batch_of(1)
    # size the sample on a 1-record probe, then pick the thread count AT THE TIMED BATCH (torch eager does not scale to hundreds of threads, and
    # what a 1-record step likes is not what a multi-record step likes): 16 / 32 / 64 / all available cores, one step each
    cands = sorted({min(avail, c) for c in (16, 32, 64, avail)})
    torch.set_num_threads(cands[min(1, len(cands) - 1)])
    tr.step(x1, y1)
    t0 = time.perf_counter()
    tr.step(x1, y1)
    t1 = time.perf_counter() - t0
    b = max(1, min(16, int(seconds_budget / 10.0 / max(t1, 1e-3))))   # ~ len(cands) probe steps + a warm-up + >= 2 timed steps inside the budget
    x, y = batch_of(b)
    tr.step(x, y)  # warm-up at the timed shape
    probe_s, skipped = {}, []
    for t in cands:
        # ascending; once a step takes more than twice the best so far, larger thread counts are not run (torch eager's intra-op pool gets slower, not
        # faster, past the point where it stops scaling: 256 threads took 185 s for a step that 16 threads do in 0.8 s on a 2 x 64-core EPYC 9575F)
        if probe_s and min(probe_s.values()) * 2.0 < list(probe_s.values())[-1]:
            skipped.append(t)
            continue
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        tr.step(x, y)
        probe_s[t] = time.perf_counter() - t0
    cores = min(probe_s, key=probe_s.get)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < seconds_budget * 0.4 and n < 8):
        tr.step(x, y)
        n += 1
    dt = time.perf_counter() - t0
    probes = ', '.join(f'{t}: {b / v:.2f} rec/s' for t, v in probe_s.items())
    if skipped:
        probes += f'; {"/".join(str(t) for t in skipped)} (= all cores) not run: throughput had already fallen to less than half of the best'
    return dict(value=b * n / dt, unit='records/s', cores=cores, kind='port', cpu_model=cpu_model_name(), cores_available=avail,
                sample=f'{n} full train steps (fwd+BCE+bwd+clip+AdamW, torch eager f32) of the same model on {b} synthetic '
                       f'12x{conf.max_signal_length} records, {cores} threads of {avail} available (fastest of one {b}-record step each at threads {probes})',
                thread_probe_records_per_s={str(t): b / v for t, v in probe_s.items()},
                tflops=b * n * flops_rec / dt / 1e12)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='base', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'fp8'], help="fp8 = bf16 path with e4m3/e5m2 operands in the block Linears (fp8 MFMA)")
    ap.add_argument('--patch', type=int, default=20)
    ap.add_argument('--length', type=int, default=5000)
    ap.add_argument('--dropout', type=float, default=None, help='override (default: reference config default 0.1)')
    ap.add_argument('--objective', default='supervised', choices=['supervised', 'masked'],
                    help="'supervised' = the reference's BCE step; 'masked' = the build's SimMIM-style masked pre-train step (seq = n patches, no CLS)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-probe', action='store_true')
    ap.add_argument('--no-masked', action='store_true', help='skip the nested masked pre-train measurement')
    ap.add_argument('--no-bf16-saved', action='store_true', help='skip the nested measurement of the same step with the saved FFN tensor in bf16')
    ap.add_argument('--no-small', action='store_true', help='skip the nested EcgVit-small measurement (BASELINE.json configs[1])')
    ap.add_argument('--no-fp8-large', action='store_true', help='skip the nested EcgVit-large fp8 / bf16 measurement (BASELINE.json configs[4] on one GPU)')
    ap.add_argument('--defer-nonfinite', action='store_true', help="read the optimiser's non-finite flag one step late (no per-step host sync)")
    ap.add_argument('--single-rank-collectives', action='store_true',
                    help='diagnostic, --gpus 1 only: a 1-rank RCCL group with the N > 1 code path switched on (self-launcher, start broadcast, '
                         'per-bucket all-reduces overlapped with backward, chunked GEMM launches): what the data-parallel machinery costs without a wire')
    ap.add_argument('--grad-comm', choices=['f32', 'bf16'], default='f32', help='dtype of the gradient buckets on the wire (N > 1)')
    ap.add_argument('--hip-lib', default=None, help='measurement only: load another build of the same C-ABI (A/B candidate, tools build)')
    ap.add_argument('--mask-on-device', action='store_true', help='measurement only (A/B): hand the masked step device-resident indices (validated with three blocking reads per step)')
    ap.add_argument('--bf16-aux', action='store_true', help='measurement only (A/B): the saved FFN tensor as bf16 (round 4) instead of e4m3 bytes')
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same command (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment -- what torch.distributed.run would set), relay rank 0's stdout, fail if any rank fails.
    Runs BEFORE anything in this process touches the GPU, starts children (never replaces this process image) and returns the exit code.
    Watchdog: every child is polled; the first non-zero exit terminates (then kills) the others at once -- a rank that dies at import,
    on an EINVAL or out of memory must not leave its peers inside an RCCL rendezvous or collective until the 10-minute NCCL timeout."""
    import subprocess
    import threading
    n = args.gpus
    env = dict(os.environ)
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # rank 0's pipe never fills while we poll
    reader.start()
    rcs = [None] * n
    failed_first = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and failed_first is None:
            failed_first = bad[0]
            for r, p in enumerate(procs):            # exactly the processes this parent started, by handle
                if rcs[r] is None:
                    p.terminate()
            deadline = time.monotonic() + 5.0
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=max(0.1, deadline - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    sys.stdout.write((out0[0] if out0 else b'').decode(errors='replace'))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        first = f'; rank {failed_first} failed first, the others were terminated' if failed_first is not None and len(bad) > 1 else ''
        print(f'bench.py: ranks failed (rank, exit code): {bad}{first}', file=sys.stderr)
        return 1
    return 0


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.single_rank_collectives):
        sys.exit(self_launch(args))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch N ranks, or run without a launcher)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or args.single_rank_collectives:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        rccl_ranks = dist.get_world_size()
    else:
        rccl_ranks = 0

    if args.hip_lib:
        from ecg_representation_learning_amd import hip as _hip
        _hip.use_library(args.hip_lib)
    import ecg_representation_learning_amd as E
    E.hip.lib()  # fail loudly if the HIP library is missing

    conf, batch = make_config(E, args.config, args.patch, args.length, args.dropout)
    batch = args.batch or batch
    dtype = torch.float32 if args.dtype == 'f32' else torch.bfloat16
    fp8 = args.dtype == 'fp8'

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(objective, steps, warmup, conf=conf, batch=batch, dtype=dtype, fp8=fp8, saved_e4m3=None):
        """W untimed + K timed steps of one objective; returns (seconds (max over ranks), final loss, probe result, ms per step of every rank)"""
        torch.manual_seed(77)  # identical initial weights on every rank (HipTrainStep broadcasts rank 0's anyway)
        if args.bf16_aux:
            saved_e4m3 = False
        model = E.EcgVit(config=conf, compute_dtype=dtype, fp8_linear=fp8, saved_ffn_e4m3=saved_e4m3)
        if objective == 'masked':
            model = E.MaskedEcgVit(model, mask_ratio=0.5)
        model = model.to(dev).train()
        x, y = E.workload.synthetic_batch(batch, length=conf.max_signal_length, seed=77 + rank)
        x, y = x.to(dev), y.to(dev)
        if objective == 'masked':
            y = model.random_mask_indices(batch, generator=torch.Generator().manual_seed(77 + rank))   # (B, m) int32 on the HOST, as a loop that draws its masks per step has them: validated there, copied per step
            if args.mask_on_device:
                y = y.to(dev)
        n_total = steps + warmup
        step = E.HipTrainStep(model, E.get_train_args(dict(train_batch_size=batch * world, num_train_epoch=1), n_train=batch * world * n_total),
                              sync_nonfinite=not args.defer_nonfinite, single_rank_collectives=args.single_rank_collectives,
                              grad_comm_dtype=torch.bfloat16 if args.grad_comm == 'bf16' else torch.float32)
        run_step = step.step_masked if objective == 'masked' else step.step
        probe = None
        if not args.no_probe and dtype == torch.bfloat16:
            probe = GemmProbe(E.hip, E.hip.GEMM_NT, torch.bfloat16)
            probe.install()
        for _ in range(warmup):
            run_step(x, y)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            if probe:
                probe.enabled = rank == 0 and i % 4 == 0   # HIP events around the dominant kernel's launches on every 4th timed step
            loss, _ = run_step(x, y)                      # (192 event records per probed step cost ~0.7 % when taken on every step)
        sync()
        dt = time.perf_counter() - t0
        if probe:
            probe.enabled = False
            probe.uninstall()
        rank_dt = [dt]
        if world > 1 or args.single_rank_collectives:
            import torch.distributed as dist
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            rank_dt = [float(e.item()) for e in every]
            dt = max(rank_dt)          # MAX over ranks
        final_loss = float(loss)
        step.finish()
        return dt, final_loss, (probe.result() if probe else None), [1e3 * d / steps for d in rank_dt]

    def roofline_of(r, objective, **over):
        if not r:
            return None
        margs = argparse.Namespace(**{**vars(args), 'objective': objective, **over})
        traffic, source = pmc_traffic('gemm_nt', margs)
        peak = PEAK_FP8_TFLOPS if r['launches_8bit'] else PEAK_BF16_TFLOPS
        return {
            'kernel': 'gemm_nt_kernel<bf16 out> (persistent quadrant-phased 256x256x64 LDS-DMA GEMM with a register-direct epilogue, A . B^T: '
                      'the Linear forward launches QKV / attn-out / FFN-up / FFN-down and, against the transposed weight shadows, their input-gradient launches'
                      + ('; e4m3 / e5m2 operands on the block-scaled fp8 MFMA, K-tile 128 deep' if r['launches_8bit'] else '') + ')',
            'bound': 'mfma', 'achieved': r['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': r['tflops'] / peak,
            'traffic': traffic, 'traffic_source': source, 'avg_launch_us': r['avg_us'], 'launches': r['launches'],
            'alg_flops_per_launch': r['flops_per_launch'], 'alg_bytes_per_launch': r['alg_bytes_per_launch'], 'bytes_model': BYTES_MODEL,
        }

    dt, final_loss, pres, rank_ms = timed_run(args.objective, args.steps, args.warmup)   # the headline run: its own per-rank times
    masked_line = None
    if args.objective == 'supervised' and world == 1 and args.config == 'base' and not args.no_masked:
        # the build's own masked pre-train objective (absent from the reference: SURVEY 0), driver-timed next to the headline step
        msteps = min(args.steps, 20)
        mdt, mloss, mres, _ = timed_run('masked', msteps, min(args.warmup, 3))
        n_patch = conf.max_signal_length // conf.patch_size
        masked_line = {
            'workload': f'EcgVit-{args.config} masked-patch pre-train step (SimMIM-style: 50 % of the {n_patch} patches replaced by a mask token, no CLS, '
                        f'L1 reconstruction of the masked patches; fwd+loss+bwd+clip+AdamW), {dropout_desc(conf)}, {batch} records/GPU',
            'value': batch * msteps / mdt, 'unit': 'records/s', 'steps': msteps, 'ms_per_step': 1e3 * mdt / msteps, 'final_loss': mloss,
            # at its OWN algorithmic FLOP count: 250 tokens (no CLS row), + the pixel head over the masked rows, no classification head
            'model_tflops_per_gpu': batch * msteps / mdt * E.workload.masked_train_flops_per_record(conf, 0.5) / 1e12,
            'mfma_frac_of_peak': batch * msteps / mdt * E.workload.masked_train_flops_per_record(conf, 0.5) / 1e12 / PEAK_BF16_TFLOPS,
            'roofline': roofline_of(mres, 'masked'),
        }
    bf16_saved_line = None
    if args.objective == 'supervised' and world == 1 and args.config == 'base' and args.dtype == 'bf16' and not args.no_bf16_saved and not args.bf16_aux:
        # the headline step keeps ONE backward-only tensor per layer (gelu'(pre) x dropout multiplier, [tokens, 3072]) as e4m3 bytes; this is the same
        # step with that tensor in bf16 -- every stored tensor then has the metric's dtype -- timed by the same caller, same steps
        bdt, bloss, _, _ = timed_run('supervised', args.steps, args.warmup, saved_e4m3=False)
        bf16_saved_line = {
            'workload': 'the headline step with saved_ffn_e4m3=False: the FFN backward tensor gelu\'(pre) x dropout multiplier stored as bf16 (2 bytes) instead of e4m3 (1 byte)',
            'value': batch * args.steps / bdt, 'unit': 'records/s', 'steps': args.steps, 'ms_per_step': 1e3 * bdt / args.steps, 'final_loss': bloss,
            'model_tflops_per_gpu': batch * args.steps / bdt * E.workload.train_flops_per_record(conf) / 1e12,
            'mfma_frac_of_peak': batch * args.steps / bdt * E.workload.train_flops_per_record(conf) / 1e12 / PEAK_BF16_TFLOPS,
        }

    small_line = None
    if args.objective == 'supervised' and world == 1 and args.config == 'base' and args.dtype == 'bf16' and not args.no_small:
        # BASELINE.json configs[1]: EcgVit-small, 250 patches (251 tokens), 256 records, driver-timed next to the headline step
        sconf, sbatch = make_config(E, 'small', args.patch, args.length, args.dropout)
        ssteps = min(args.steps, 20)
        sdt, sloss, sres, _ = timed_run('supervised', ssteps, min(args.warmup, 5), conf=sconf, batch=sbatch, dtype=torch.bfloat16, fp8=False)
        sflops = E.workload.train_flops_per_record(sconf)
        sv = sbatch * ssteps / sdt
        small_line = {
            'workload': f'EcgVit-small supervised BCE train step (fwd+loss+bwd+clip+AdamW), bf16, {dropout_desc(sconf)}, {sbatch} records/GPU x 12 leads x '
                        f'{sconf.max_signal_length} samples, patch {sconf.patch_size} ({sconf.max_signal_length // sconf.patch_size + 1} tokens)',
            'value': sv, 'unit': 'records/s', 'steps': ssteps, 'ms_per_step': 1e3 * sdt / ssteps, 'final_loss': sloss,
            'model_tflops_per_gpu': sv * sflops / 1e12, 'mfma_frac_of_peak': sv * sflops / 1e12 / PEAK_BF16_TFLOPS,
            'roofline': roofline_of(sres, 'supervised', config='small', batch=sbatch),
        }

    fp8_line = None
    if args.objective == 'supervised' and world == 1 and args.config == 'base' and args.dtype == 'bf16' and not args.no_fp8_large:
        # BASELINE.json configs[4] on one GPU, driver-timed next to the headline step: EcgVit-large, patch 10 (501 tokens), 256 records,
        # fp8 (e4m3 / e5m2) Linear operands -- and the same step with bf16 operands, back to back on this device, for the ratio
        import gc
        lconf, lbatch = make_config(E, 'large', 10, args.length, args.dropout)
        lsteps, lwarm = min(args.steps, 6), min(args.warmup, 2)
        res = {}
        for tag, f8 in (('bf16', False), ('fp8', True)):
            gc.collect()
            torch.cuda.empty_cache()
            ldt, lloss, lres, _ = timed_run('supervised', lsteps, lwarm, conf=lconf, batch=lbatch, dtype=torch.bfloat16, fp8=f8)
            res[tag] = (ldt, lloss, lres)
        gc.collect()
        torch.cuda.empty_cache()
        lflops = E.workload.train_flops_per_record(lconf)
        v8, v16 = lbatch * lsteps / res['fp8'][0], lbatch * lsteps / res['bf16'][0]
        fp8_line = {
            'workload': f'EcgVit-large supervised BCE train step, {FP8_DESC}; patch 10 ({lconf.max_signal_length // 10 + 1} tokens), '
                        f'{dropout_desc(lconf)}, {lbatch} records/GPU',
            'value': v8, 'unit': 'records/s', 'steps': lsteps, 'ms_per_step': 1e3 * res['fp8'][0] / lsteps, 'final_loss': res['fp8'][1],
            'bf16_same_config': {'value': v16, 'ms_per_step': 1e3 * res['bf16'][0] / lsteps, 'final_loss': res['bf16'][1]},
            'fp8_over_bf16': v8 / v16,
            'model_tflops_per_gpu': v8 * lflops / 1e12, 'frac_of_fp8_peak': v8 * lflops / 1e12 / PEAK_FP8_TFLOPS,
            'roofline': roofline_of(res['fp8'][2], 'supervised', config='large', dtype='fp8', patch=10, batch=lbatch),
        }

    if rank == 0:
        flops_rec = E.workload.masked_train_flops_per_record(conf, 0.5) if args.objective == 'masked' else E.workload.train_flops_per_record(conf)
        value = batch * world * args.steps / dt
        storage_desc = ''
        if dtype == torch.bfloat16:
            e4 = not args.bf16_aux
            storage_desc = ('; arithmetic bf16 MFMA / f32 accumulate, all activations stored bf16 EXCEPT one backward-only tensor per layer: the FFN\'s gelu\'(pre) x dropout '
                            'multiplier [tokens, ffn] is kept between forward and backward as ' + ('e4m3 BYTES (3 mantissa bits, relative error <= 2^-4 per element, unbiased; forward '
                            'values unaffected; the same step with it in bf16 is the nested bf16_saved_tensor object)' if e4 else 'bf16 (--bf16-aux)') +
                            '; GELU / GELU\' of the bf16 path use a three-term erf (|error| <= 2.5e-5, below one bf16 ulp of the stored values)')
        out = {
            'metric': '12-lead ECG records/sec pre-train step, ViT-Base bf16 @ 1/2/4/8 MI355X' if args.config == 'base'
            else f'12-lead ECG records/sec train step, EcgVit-{args.config}',
            'value': value, 'unit': 'records/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype if not fp8 else f'fp8 ({FP8_DESC})',
            'data': 'synthetic',
            'config': {
                'workload': f'EcgVit-{args.config} ' + ('masked-patch pre-train step (SimMIM-style, 50 % of patches masked, L1 recon; fwd+loss+bwd+clip+AdamW'
                            if args.objective == 'masked' else 'supervised BCE train step (reference train.py:271-283: fwd+loss+bwd+clip+AdamW') + (
                            f'{"+RCCL all-reduce" if world > 1 else ""}), hidden {conf.hidden_size} x {conf.num_hidden_layers} layers x {conf.num_attention_heads} heads, '
                            f'ffn {conf.intermediate_size}, {dropout_desc(conf)}, '
                            f'{batch} records/GPU x 12 leads x {conf.max_signal_length} samples, patch {conf.patch_size} '
                            f'({conf.max_signal_length // conf.patch_size + (0 if args.objective == "masked" else 1)} tokens), random-init weights, inputs resident in HBM'
                            + storage_desc),
                'global_batch': batch * world, 'per_gpu_batch': batch, 'parallelism': f'dp{world}' + ('+single-rank-collectives' if args.single_rank_collectives else ''),
                # what holds the timed path to the reference (tests/, -m gpu): stated next to the number it qualifies
                'parity': 'f32 HIP path vs CPU oracle <= 1e-4 (loss, logits, every gradient; full-depth base, masked step at this geometry); bf16 path: loss <= 2e-2, '
                          'whole-gradient cosine >= 0.98, every tensor >= 0.95 -- at dropout 0 AND at dropout 0.1 as timed (tests/test_gpu_dropout_parity.py: the masks the HIP '
                          'kernels drew are exported and injected into the oracle\'s five nn.Dropout sites, supervised and masked step, base and large layer shapes, one full-depth base model in f32, e4m3 saved tensor and '
                          'quad 8-bit masks on); the oracle\'s transformer arithmetic restates vit-pytorch 0.33.2 (not installable here: parity unpinned)',
            },
            'final_loss': final_loss,
            'workload_key': workload_key(args), 'kernel_source_sha16': kernel_source_hash(),
            'rccl_ranks': rccl_ranks,   # size of the RCCL process group the step exchanged gradients over (0 = no group: plain single-GPU step)
            'rank_ms_per_step': rank_ms,   # the headline run's own step time on every rank (max = ms_per_step)
            'model_tflops_per_gpu': value / world * flops_rec / 1e12,
            'mfma_frac_of_peak': value / world * flops_rec / 1e12 / (PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS),
        }
        if pres:
            out['roofline'] = roofline_of(pres, args.objective)
        if bf16_saved_line:
            out['bf16_saved_tensor'] = bf16_saved_line
        if masked_line:
            out['masked'] = masked_line
        if small_line:
            out['small'] = small_line
        if fp8_line:
            out['fp8_large'] = fp8_line
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(conf, masked=args.objective == 'masked')
        print(json.dumps(out), flush=True)
    if world > 1 or args.single_rank_collectives:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
