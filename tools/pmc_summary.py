#!/usr/bin/env python3
"""Fold the rocprofv3 passes of tools/pmc_bench.sh into one JSON: per kernel family the average duration (kernel trace pass), HBM bytes
per launch (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 calibration of guides/MI355X_MICROARCH.md: FETCH_SIZE counts half of wide
streaming reads; units KiB), L2 hit rate, MFMA-busy / VALU / LDS / wait counters, and the hash of the kernel sources profiled."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(n):
    if 'gemm_nt_kernel' in n: return 'gemm_nt'           # dominant symbol: persistent A.B^T GEMM (forward + input gradients)
    if 'gemm_wgrad8_kernel' in n: return 'gemm_wgrad8'   # the same on 8-bit operands (fp8_linear)
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
    out_dir, tag, bench_args = sys.argv[1], sys.argv[2], sys.argv[3:]
    import bench
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(out_dir, '*', '*', '*counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            k = family(r['Kernel_Name'])
            if k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    traces = sorted(glob.glob(os.path.join(out_dir, 'trace', '*', '*kernel_trace.csv')))
    assert len(traces) == 1, f'expected exactly one trace pass under {out_dir}/trace, found {traces}'
    for r in csv.DictReader(open(traces[0])):
        k = family(r['Kernel_Name'])
        if k:
            dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-3)
    # the --stats table of THE SAME trace pass travels with the summary (tools/check_profiles.py recomputes the averages from it)
    stats = traces[0].replace('kernel_trace.csv', 'kernel_stats.csv')
    shutil.copyfile(stats, os.path.join(out_dir, 'kernel_stats.csv'))
    stat_fam = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats)):
        k = family(r['Name'])
        if k:
            stat_fam[k][0] += int(r['Calls'])
            stat_fam[k][1] += float(r['TotalDurationNs'])
    kernels = {}
    for k in sorted(set(agg) | set(dur)):
        o = {c: sum(v) / len(v) for c, v in agg[k].items()}
        if dur[k]:
            o['launches_per_run'] = len(dur[k])
            o['avg_us'] = sum(dur[k]) / len(dur[k])
            o['total_ms'] = sum(dur[k]) * 1e-3
        if k in stat_fam:
            o['stats_calls'], o['stats_total_ns'] = stat_fam[k]
        if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o:
            o['hbm_bytes_per_launch'] = (2.0 * o['FETCH_SIZE'] + o['WRITE_SIZE']) * 1024.0
            if 'avg_us' in o:
                o['hbm_GBps'] = o['hbm_bytes_per_launch'] / o['avg_us'] * 1e-3
        if 'TCC_HIT_sum' in o:
            o['l2_hit_rate'] = o['TCC_HIT_sum'] / max(1.0, o['TCC_HIT_sum'] + o['TCC_MISS_sum'])
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in o and o.get('SQ_BUSY_CYCLES'):
            o['mfma_busy_over_sq_busy'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in o and o.get('GRBM_GUI_ACTIVE'):
            # MFMA-busy share of the SIMD cycles: busy cycles summed over 1024 SIMDs / (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)
            o['mfma_busy_frac'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (o['GRBM_GUI_ACTIVE'] / 8.0)
        kernels[k] = o
    wargs = bench.parse_args(bench_args)
    res = dict(tag=tag, workload_key=bench.workload_key(wargs), command='bench.py ' + ' '.join(bench_args),
               kernel_source_sha16=bench.kernel_source_hash(), steps_in_run=wargs.steps + wargs.warmup,
               stats_csv='kernel_stats.csv (the --stats table of the same trace pass; stats_calls / stats_total_ns per family are read from it)',
               notes='per-launch averages over all launches of a family in the run (steps + warm-up); FETCH_SIZE/WRITE_SIZE in KiB; '
                     'hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of wide streaming reads); '
                     'SQ_* summed over all SEs/XCDs; profiled passes run 2-3 % slower than unprofiled ones (DVFS), durations come from the trace-only pass',
               kernels=kernels)
    with open(os.path.join(out_dir, 'summary.json'), 'w') as f:
        json.dump(res, f, indent=1)
    for k, o in kernels.items():
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in o.items() if a in ('avg_us', 'launches_per_run', 'hbm_bytes_per_launch', 'hbm_GBps', 'l2_hit_rate', 'mfma_busy_over_sq_busy', 'mfma_busy_frac', 'total_ms')})


if __name__ == '__main__':
    main()
