#!/usr/bin/env python3
"""Fold the rocprofv3 passes of tools/pmc_bench.sh into one JSON: per kernel family the average duration (kernel trace pass), HBM bytes
per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 calibration of guides/MI355X_MICROARCH.md: FETCH_SIZE counts half of wide
streaming reads; units KiB), L2 hit rate, MFMA-busy / VALU / LDS / wait counters, and the hash of the kernel sources profiled."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(n):
    if 'gemm_nt_kernel' in n: return 'gemm_nt'           # dominant symbol: persistent A.B^T GEMM (forward + input gradients)
    if 'gemm_wgrad_kernel' in n: return 'gemm_wgrad'     # streaming weight-gradient GEMM
    if 'splitk_reduce' in n: return 'splitk_reduce'
    if 'gemm_bf16_kernel' in n: return 'gemm_128'        # small / ragged products
    if 'attn_fwd' in n: return 'attn_fwd'
    if 'attn_bwd' in n: return 'attn_bwd'
    if 'layernorm_bwd' in n: return 'ln_bwd'
    if 'layernorm_fwd' in n: return 'ln_fwd'
    if 'adamw' in n: return 'adamw'
    if 'transpose_batched' in n: return 'transpose'
    if 'patch_gather' in n: return 'patch_gather'
    return None


def main():
    out_dir, obj = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(out_dir, '*', '*', '*counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            k = family(r['Kernel_Name'])
            if k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out_dir, 'trace', '*', '*kernel_trace.csv'))):
        for r in csv.DictReader(open(f)):
            k = family(r['Kernel_Name'])
            if k:
                dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    kernels = {}
    for k in sorted(set(agg) | set(dur)):
        o = {c: sum(v) / len(v) for c, v in agg[k].items()}
        if dur[k]:
            o['launches_per_run'] = len(dur[k])
            o['avg_us'] = sum(dur[k]) / len(dur[k])
            o['total_ms'] = sum(dur[k]) * 1e-3
        if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o:
            o['hbm_bytes_per_launch'] = (2.0 * o['FETCH_SIZE'] + o['WRITE_SIZE']) * 1024.0
            if 'avg_us' in o:
                o['hbm_GBps'] = o['hbm_bytes_per_launch'] / o['avg_us'] * 1e-3
        if 'TCC_HIT_sum' in o:
            o['l2_hit_rate'] = o['TCC_HIT_sum'] / max(1.0, o['TCC_HIT_sum'] + o['TCC_MISS_sum'])
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in o and o.get('SQ_BUSY_CYCLES'):
            o['mfma_busy_over_sq_busy'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES']
        kernels[k] = o
    import bench
    res = dict(objective=obj, command=f'bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-masked --objective {obj} (EcgVit-base bf16, 512 x 12 x 5000)',
               kernel_source_sha16=bench.kernel_source_hash(),
               notes='per-launch averages over all launches of a family in the run (3 steps incl. warm-up); FETCH_SIZE/WRITE_SIZE in KiB; '
                     'hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide streaming reads); '
                     'SQ_* summed over all SEs/XCDs; profiled passes run 2-3 % slower than unprofiled ones (DVFS), durations come from the trace-only pass',
               kernels=kernels)
    with open(os.path.join(out_dir, 'summary.json'), 'w') as f:
        json.dump(res, f, indent=1)
    for k, o in kernels.items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in o.items() if a in ('avg_us', 'launches_per_run', 'hbm_bytes_per_launch', 'hbm_GBps', 'l2_hit_rate', 'mfma_busy_over_sq_busy', 'total_ms')})


if __name__ == '__main__':
    main()
