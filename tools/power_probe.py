#!/usr/bin/env python3
"""Board power and shader clock while one gemm_nt launch shape loops (rocm-smi sampled from a side thread):
is the chip power-capped under this kernel?   usage: python tools/power_probe.py [seconds]"""
import ctypes
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ecg_representation_learning_amd as E  # noqa: E402,F401
from ecg_representation_learning_amd import hip  # noqa: E402
from ecg_representation_learning_amd.hip import GEMM_NT  # noqa: E402


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
    except Exception as ex:  # noqa: BLE001
        return repr(ex)
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('Power', 'sclk', 'fclk', 'mclk', 'Temperature (Sensor junction)', 'edge'))]
    return ' | '.join(keep)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    print('idle:', smi(), flush=True)
    M, bf, dev = 512 * 251, torch.bfloat16, 'cuda'
    for name, K, N in (('qkv', 768, 2304), ('ffn_down', 3072, 768)):
        X = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        for which in ('gemm_nt', 'torch.matmul'):
            stop = False
            samples = []

            def sampler():
                time.sleep(1.0)
                while not stop:
                    samples.append(smi())
                    time.sleep(0.5)
            th = threading.Thread(target=sampler)
            th.start()
            t0 = time.time()
            n = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            while time.time() - t0 < secs:
                for _ in range(50):
                    if which == 'gemm_nt':
                        hip.gemm(GEMM_NT, X, W, C, M, N, K, K, K, N)
                    else:
                        torch.matmul(X, W.t(), out=C)
                n += 50
                torch.cuda.synchronize()
            e1.record()
            torch.cuda.synchronize()
            stop = True
            th.join()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(f'{name} {which}: {us:7.1f} us per launch, {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s', flush=True)
            for s in samples[:4]:
                print('    ', s, flush=True)


if __name__ == '__main__':
    main()
