#!/usr/bin/env python3
"""Evidence-chain check over profiles/ (CPU; run by tests/test_profiles.py).

For every counter summary `profiles/rNN_pmc_<tag>.json` written by tools/pmc_bench.sh (round 3 on: it names its `stats_csv`):
  * `profiles/rNN_pmc_<tag>_kernel_stats.csv` must exist and be the `--stats` table of the SAME trace pass: per kernel family its calls and
    total nanoseconds must equal the summary's `stats_calls` / `stats_total_ns`, and calls == `launches_per_run`, total == sum of the
    traced durations (0.1 %).
For every bench line `profiles/rNN_bench_line*.json` that carries `kernel_source_sha16` and a `roofline`:
  * a summary with the same `workload_key` and the same source hash must exist, and the roofline fraction recomputed from ITS stats CSV
    (algorithmic FLOPs per launch of the bench line / average nanoseconds of the gemm_nt family / peak) must agree with the bench line's
    `roofline.frac` within 3 % (a profiled pass clocks 2-3 % lower than an unprofiled one: guides/MI355X_MICROARCH.md, DVFS give-back 2).
For every text table `profiles/rNN_*.txt` of round 4 on that names the sources it was taken on (a header line `kernel_source_sha16: <hash>`):
  * the hash must be the one of the newest bench line of the same round (`profiles/rNN_bench_line.json`): a table left over from an earlier
    build of the round fails the check instead of passing as "final" (round 3's r03_gemm_shapes.txt did exactly that).
Exit code 0 = consistent; prints one line per check."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from pmc_summary import family  # noqa: E402


def stats_by_family(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = family(r['Name'])
            if k:
                c, t = out.get(k, (0, 0.0))
                out[k] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    return out


def check(profiles=None, verbose=True):
    profiles = profiles or os.path.join(ROOT, 'profiles')
    errors, checked = [], 0
    summaries = {}
    for p in sorted(glob.glob(os.path.join(profiles, 'r*_pmc_*.json'))):
        with open(p) as f:
            s = json.load(f)
        if 'stats_csv' not in s:          # rounds 1-2: no same-pass CSV was kept (profiles/README.md says so)
            continue
        name = os.path.basename(p)
        c = p[:-5] + '_kernel_stats.csv'
        if not os.path.exists(c):
            errors.append(f'{name}: {os.path.basename(c)} missing')
            continue
        st = stats_by_family(c)
        for k, o in s['kernels'].items():
            if 'stats_calls' not in o:
                continue
            checked += 1
            calls, tot = st.get(k, (0, 0.0))
            if calls != o['stats_calls'] or abs(tot - o['stats_total_ns']) > 0.5:
                errors.append(f'{name}: {k}: CSV has {calls} calls / {tot:.0f} ns, summary recorded {o["stats_calls"]} / {o["stats_total_ns"]:.0f} (different trace pass)')
            elif calls != o.get('launches_per_run') or abs(tot - 1e6 * o.get('total_ms', 0)) > 1e-3 * tot:
                errors.append(f'{name}: {k}: stats table ({calls} calls, {tot / 1e6:.3f} ms) disagrees with the traced launches ({o.get("launches_per_run")}, {o.get("total_ms")} ms)')
        summaries[(s.get('workload_key'), s.get('kernel_source_sha16'))] = (name, s, st)
        if verbose:
            print(f'ok  {name}: {len(s["kernels"])} kernel families agree with {os.path.basename(c)}')
    for p in sorted(glob.glob(os.path.join(profiles, 'r*_bench_line*.json'))):
        with open(p) as f:
            text = f.read()
        try:   # round-1 files hold the raw stdout of the run (a driver warning line in front of the JSON line)
            b = json.loads(text[text.index('{'):])
        except ValueError:
            continue
        roof, sha, key = b.get('roofline'), b.get('kernel_source_sha16'), b.get('workload_key')
        if not roof or not sha:
            continue
        name = os.path.basename(p)
        hit = summaries.get((key, sha))
        if hit is None:
            errors.append(f'{name}: no counter summary of workload {key} on sources {sha} under profiles/ (found: {sorted(k for k in summaries)})')
            continue
        sname, s, st = hit
        calls, tot = st.get('gemm_nt', (0, 0.0))
        if not calls:
            errors.append(f'{name}: {sname} has no gemm_nt launches')
            continue
        checked += 1
        frac = roof['alg_flops_per_launch'] / (tot / calls * 1e-9) / (roof['peak'] * 1e12)
        rel = abs(frac - roof['frac']) / roof['frac']
        if rel > 0.03:
            errors.append(f'{name}: roofline.frac {roof["frac"]:.4f} vs {frac:.4f} recomputed from {sname}\'s stats CSV ({calls} launches, {tot / calls / 1e3:.1f} us avg): {100 * rel:.1f} % apart')
        elif verbose:
            print(f'ok  {name}: roofline.frac {roof["frac"]:.4f} vs {frac:.4f} from {sname} ({tot / calls / 1e3:.1f} us avg over {calls} launches)')
        if roof.get('traffic') is not None and verbose:
            # traffic / algorithmic bytes is only comparable between lines that count the same streams: bench.BYTES_MODEL (absent: rounds <= 4 counted
            # operands + output = model 1; round 5's line already counted the epilogue streams = model 2 without saying so)
            bm = roof.get('bytes_model', 2 if name.startswith('r05') else 1)
            print(f'    {name}: traffic / algorithmic bytes = {roof["traffic"] / roof["alg_bytes_per_launch"]:.2f} (bytes model {bm}; compare only with lines of the same model)')
        if roof.get('traffic') is not None and abs(roof['traffic'] - s['kernels']['gemm_nt'].get('hbm_bytes_per_launch', -1)) > 1:
            errors.append(f'{name}: roofline.traffic {roof["traffic"]} is not {sname}\'s gemm_nt hbm_bytes_per_launch')
    # text tables that name their sources: same hash as the round's bench line
    for p in sorted(glob.glob(os.path.join(profiles, 'r*_*.txt'))):
        name = os.path.basename(p)
        m = re.match(r'r(\d+)_', name)
        if not m or int(m.group(1)) < 4:
            continue
        with open(p) as f:
            head = f.read(4096)
        hs = re.findall(r'kernel_source_sha16:\s*([0-9a-f]{16})', head)
        if not hs:
            continue
        line = os.path.join(profiles, f'r{m.group(1)}_bench_line.json')
        if not os.path.exists(line):
            errors.append(f'{name}: names sources {hs[0]} but profiles/{os.path.basename(line)} does not exist')
            continue
        with open(line) as f:
            text = f.read()
        want = json.loads(text[text.index('{'):]).get('kernel_source_sha16')
        checked += 1
        if any(h != want for h in hs):
            errors.append(f'{name}: taken on sources {hs[0]}, the round\'s bench line is on {want} (stale table)')
        elif verbose:
            print(f'ok  {name}: same sources as {os.path.basename(line)} ({want})')
    return errors, checked


if __name__ == '__main__':
    errs, n = check()
    for e in errs:
        print('ERR', e)
    print(f'{n} checks, {len(errs)} errors')
    sys.exit(1 if errs else 0)
