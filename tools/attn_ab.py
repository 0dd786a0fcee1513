#!/usr/bin/env python3
"""A/B timing of the fused attention kernels of TWO builds of the library in ONE process on ONE device (interleaved rounds), at the
benchmark shape (512 records x 12 heads x 251 tokens, dropout 0.1) -- and a bitwise comparison of what the two builds produce.
usage: python tools/attn_ab.py [libA.so libB.so] [--n 251] [--p 0.1]   (defaults: csrc/build/libecgvit_hip_prev.so vs the shipped library)"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecg_representation_learning_amd import hip  # noqa: E402


import bench as _bench  # noqa: E402
print('kernel_source_sha16:', _bench.kernel_source_hash(), '(sources of the library build measured: tools/check_profiles.py holds committed tables to the round\'s bench line)', flush=True)
def load(path):
    l = ctypes.CDLL(path)
    for name in ('ecgvit_attention_fwd', 'ecgvit_attention_bwd'):
        fn = getattr(l, name)
        fn.restype, fn.argtypes = hip.SIGNATURES[name]
    return l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='*')
    ap.add_argument('--b', type=int, default=512)
    ap.add_argument('--n', type=int, default=251)
    ap.add_argument('--h', type=int, default=12)
    ap.add_argument('--p', type=float, default=0.1)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=12)
    a = ap.parse_args()
    pk = os.path.join(ROOT, 'ecg-representation-learning_amd')
    paths = a.libs or [os.path.join(pk, 'csrc', 'build', 'libecgvit_hip_prev.so'), os.path.join(pk, 'libecgvit_hip.so')]
    libs = [load(p) for p in paths]
    B, N, h, dh = a.b, a.n, a.h, 64
    d, bf = h * dh, torch.bfloat16
    torch.manual_seed(3)
    qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf)
    do = torch.randn(B * N, d, device='cuda').to(bf)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for l in libs:
        out = torch.empty(B * N, d, device='cuda', dtype=bf)
        lse = torch.empty(B * h * N, device='cuda')
        dqkv = torch.zeros(B * N, 3 * d, device='cuda', dtype=bf)
        assert l.ecgvit_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, h, dh, 0.125, a.p, 7, hip.BF16, st) == 0
        assert l.ecgvit_attention_bwd(qkv.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, h, dh, 0.125, a.p, 7, hip.BF16, st) == 0
        torch.cuda.synchronize()
        res.append((out, lse, dqkv))
    for name, i in (('out', 0), ('lse', 1), ('dqkv', 2)):
        x, y = res[0][i], res[1][i]
        same = torch.equal(x, y)
        print(f'{name}: {"bit-identical" if same else "DIFFERENT: rel %.3e, %d elements" % (float((x.float() - y.float()).norm() / y.float().norm()), int((x != y).sum()))}')
    t = {(i, k): [] for i in range(len(libs)) for k in ('fwd', 'bwd')}
    for _ in range(a.rounds):
        for i, l in enumerate(libs):
            out, lse, dqkv = res[i]
            for k in ('fwd', 'bwd'):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    if k == 'fwd':
                        l.ecgvit_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, h, dh, 0.125, a.p, 7, hip.BF16, st)
                    else:
                        l.ecgvit_attention_bwd(qkv.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, h, dh, 0.125, a.p, 7, hip.BF16, st)
                e1.record()
                torch.cuda.synchronize()
                t[(i, k)].append(1e3 * e0.elapsed_time(e1) / a.iters)
    for i, p in enumerate(paths):
        for k in ('fwd', 'bwd'):
            v = sorted(t[(i, k)][1:])
            print(f'{os.path.basename(p):28s} {k}: median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us')


if __name__ == '__main__':
    main()
