"""CPU: the committed profiles are self-consistent (tools/check_profiles.py): every counter summary travels with the --stats CSV of its
own trace pass, and every bench line's roofline fraction can be recomputed from that CSV on the same kernel sources."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_profiles_evidence_chain():
    import check_profiles
    errors, checked = check_profiles.check(verbose=False)
    assert not errors, '\n'.join(errors)


def test_check_catches_a_stale_csv(tmp_path):
    """the failure mode of round 2: a summary next to the stats table of ANOTHER trace pass"""
    import json
    import check_profiles
    s = dict(stats_csv='kernel_stats.csv', workload_key='w', kernel_source_sha16='0' * 16,
             kernels=dict(gemm_nt=dict(stats_calls=96, stats_total_ns=44294400.0, launches_per_run=96, total_ms=44.2944, avg_us=461.4)))
    with open(tmp_path / 'r99_pmc_x.json', 'w') as f:
        json.dump(s, f)
    row = '"void (anonymous namespace)::gemm_nt_kernel<__bf16, 0>(ecgvit_gemm_desc)",96,{tot},{avg},50.0,1,2,3\n'
    head = '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
    with open(tmp_path / 'r99_pmc_x_kernel_stats.csv', 'w') as f:
        f.write(head + row.format(tot=45580800, avg=474800))          # the 474.8-us table next to a 461.4-us summary
    errors, _ = check_profiles.check(str(tmp_path), verbose=False)
    assert errors and 'different trace pass' in errors[0]
    with open(tmp_path / 'r99_pmc_x_kernel_stats.csv', 'w') as f:
        f.write(head + row.format(tot=44294400, avg=461400))
    errors, n = check_profiles.check(str(tmp_path), verbose=False)
    assert not errors and n == 1
    b = dict(kernel_source_sha16='0' * 16, workload_key='w',
             roofline=dict(alg_flops_per_launch=4.548e11, peak=2500.0, frac=0.3942, traffic=None))
    with open(tmp_path / 'r99_bench_line.json', 'w') as f:
        json.dump(b, f)
    errors, n = check_profiles.check(str(tmp_path), verbose=False)
    assert not errors and n == 2
    b['roofline']['frac'] = 0.45
    with open(tmp_path / 'r99_bench_line.json', 'w') as f:
        json.dump(b, f)
    errors, _ = check_profiles.check(str(tmp_path), verbose=False)
    assert errors and 'apart' in errors[0]


def test_check_catches_a_stale_text_table(tmp_path):
    """round 3's other failure mode: a per-shape table headed "final sources" that was taken before the final kernels"""
    import json
    import check_profiles
    with open(tmp_path / 'r04_bench_line.json', 'w') as f:
        json.dump(dict(kernel_source_sha16='a' * 16, workload_key='w'), f)
    with open(tmp_path / 'r04_gemm_shapes.txt', 'w') as f:
        f.write('per-shape table\nkernel_source_sha16: ' + 'b' * 16 + '\n')
    errors, _ = check_profiles.check(str(tmp_path), verbose=False)
    assert errors and 'stale table' in errors[0]
    with open(tmp_path / 'r04_gemm_shapes.txt', 'w') as f:
        f.write('per-shape table\nkernel_source_sha16: ' + 'a' * 16 + '\n')
    errors, n = check_profiles.check(str(tmp_path), verbose=False)
    assert not errors and n == 1
