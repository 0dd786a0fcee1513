#!/usr/bin/env python3
"""A/B of the two attention FORWARD kernels of ONE tools build in ONE process on ONE device (interleaved rounds): the one-(record, head)-per-workgroup
kernel (ecgvit_tools_attn_fwd_variant(0)) against the streamed persistent kernel of round 6 (variant 1), plain and 8-bit-emitting entry points, and a
bitwise comparison of what they produce.
usage: python tools/attn_fwd_ab.py [--b 512 --h 12 --n 251] [--p 0.1]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ecg_representation_learning_amd import hip  # noqa: E402
import toolslib  # noqa: E402
import bench as _bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=512)
    ap.add_argument('--n', type=int, default=251)
    ap.add_argument('--h', type=int, default=12)
    ap.add_argument('--p', type=float, default=0.1)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=12)
    a = ap.parse_args()
    print('kernel_source_sha16:', _bench.kernel_source_hash(), flush=True)
    tl = toolslib.tools_lib()
    B, N, h, dh = a.b, a.n, a.h, 64
    d, bf = h * dh, torch.bfloat16
    torch.manual_seed(3)
    qkv = torch.randn(B * N, 3 * d, device='cuda').to(bf)
    st = torch.cuda.current_stream().cuda_stream
    sc = torch.full((1,), 0.01, device='cuda')
    amax = torch.zeros(1, device='cuda')
    bufs = {}
    VS = (0, 1, 2) if N <= 256 else (0, 1)   # 2: the 8-wave streamed form, two workgroups per CU (N <= 256 only)
    for v in VS:
        bufs[v] = (torch.empty(B * N, d, device='cuda', dtype=bf), torch.empty(B * h * N, device='cuda'), torch.empty(B * N, d, device='cuda', dtype=torch.uint8))

    def run(v, q8):
        out, lse, o8 = bufs[v]
        tl.ecgvit_tools_attn_fwd_variant(v)
        if q8:
            rc = tl.ecgvit_attention_fwd_q8(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, h, dh, 0.125, a.p, 7, o8.data_ptr(), sc.data_ptr(), amax.data_ptr(), st)
        else:
            rc = tl.ecgvit_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, N, h, dh, 0.125, a.p, 7, hip.BF16, st)
        assert rc == 0, rc
    for q8 in (False, True):
        for v in VS:
            run(v, q8)
        torch.cuda.synchronize()
        same = all(torch.equal(bufs[0][i].view(torch.uint8), bufs[v][i].view(torch.uint8)) for v in VS[1:] for i in ((0, 1, 2) if q8 else (0, 1)))
        print(f'{B} x {h} x {N}, p = {a.p}, {"8-bit emitting" if q8 else "plain"}: one-item vs streamed outputs {"bit-identical" if same else "DIFFERENT"}')
        t = {v: [] for v in VS}
        for _ in range(a.rounds):
            for v in VS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(v, q8)
                e1.record()
                torch.cuda.synchronize()
                t[v].append(1e3 * e0.elapsed_time(e1) / a.iters)
        for v, nm in ((0, 'one item per workgroup'), (1, 'streamed (persistent, 16 waves)'), (2, 'streamed (8 waves, 2 wg / CU)'))[:len(VS)]:
            x = sorted(t[v][1:])
            print(f'   {nm:34s} median {x[len(x) // 2]:7.1f} us  min {x[0]:7.1f} us')
    tl.ecgvit_tools_attn_fwd_variant(-1)


if __name__ == '__main__':
    main()
