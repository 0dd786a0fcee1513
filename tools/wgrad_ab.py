#!/usr/bin/env python3
"""The four weight-gradient products of one EcgVit-base layer (dW = dY^T . X, f32 output, K = 512*251 token rows): gemm_wgrad_kernel_4w
(what ships) against the eight-wave gemm_wgrad_kernel (tools build: ecgvit_tools_wgrad_body) and torch.matmul (hipBLASLt, bf16 operands, f32 result via a bf16 output upcast is NOT equivalent -- the library is timed with bf16
output as a lower bound of its work), interleaved in one process.  usage: python tools/wgrad_ab.py [rounds] [records tokens hidden]   (e.g. 5 256 251 512 = the EcgVit-small shapes)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
hip.use_library(os.path.join(ROOT, 'ecg-representation-learning_amd', 'csrc', 'build', 'libecgvit_hip_tools.so'))   # explicit: the diagnostic build
from ecg_representation_learning_amd.hip import GEMM_TN  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    recs, toks, d = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (512, 251, 768)
    M, f, bf, dev = recs * toks, 4 * d, torch.bfloat16, 'cuda'
    shapes = [('qkv', 3 * d, d), ('out', d, d), ('ffn_up', f, d), ('ffn_down', d, f)]
    ws = torch.empty(hip.gemm_workspace_bytes(GEMM_TN, bf, f, f, M) + (64 << 20), dtype=torch.uint8, device=dev)
    data = {}
    for name, mo, ni in shapes:
        data[name] = ((torch.randn(M, mo, device=dev) * 0.1).to(bf), torch.randn(M, ni, device=dev).to(bf), torch.empty(mo, ni, device=dev), torch.empty(mo, ni, device=dev, dtype=bf))
    body = hip.lib().ecgvit_tools_wgrad_body
    body.restype, body.argtypes = None, [__import__('ctypes').c_int]
    res = {}
    outs = {}
    for _ in range(rounds):
        for name, mo, ni in shapes:
            dY, X, G, Gb = data[name]
            for which in ('4w', '8w', 'lib'):
                body(1 if which == '8w' else 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    if which != 'lib':
                        hip.gemm(GEMM_TN, dY, X, G, mo, ni, M, mo, ni, ni, workspace=ws)
                    else:
                        torch.matmul(dY.t(), X, out=Gb)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((name, which), []).append(e0.elapsed_time(e1) * 1e3 / 5)
                if which != 'lib':
                    outs[(name, which)] = G.clone()
    for name, mo, ni in shapes:
        for which in ('4w', '8w', 'lib'):
            t = sorted(res[(name, which)])[len(res[(name, which)]) // 2]
            print(f'{name:9s} dW {mo:4d} x {ni:4d} {which:6s}: {t:7.1f} us  {2.0 * M * mo * ni / t * 1e-6:7.1f} TFLOP/s ({2.0 * M * mo * ni / t * 1e-6 / 25:.1f} %)')
        print(f'          4w == 8w bit for bit: {bool(torch.equal(outs[(name, "4w")], outs[(name, "8w")]))}')
        body(0)
        dY, X, G, Gb = data[name]
        ref = (dY[:4096].float().t() @ X[:4096].float())
        hip.gemm(GEMM_TN, dY[:4096], X[:4096], G, mo, ni, 4096, mo, ni, ni, workspace=ws)
        print(f'          check (first 4096 rows): rel err {float((G - ref).norm() / ref.norm()):.2e}')


if __name__ == '__main__':
    main()
