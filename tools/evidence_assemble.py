#!/usr/bin/env python3
"""copy what tools/evidence_round.sh wrote (gpurun_out/evidence_<round>/) into profiles/ under the committed names, each text table behind a
prose header that says what it is and how it was taken (the `kernel_source_sha16:` line each tool printed stays: tools/check_profiles.py holds
it to the round's bench line).  usage: python tools/evidence_assemble.py r04 ["note for the step A/B"]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rn = sys.argv[1] if len(sys.argv) > 1 else 'r04'
note = sys.argv[2] if len(sys.argv) > 2 else ''
E, P = os.path.join(ROOT, 'gpurun_out', f'evidence_{rn}'), os.path.join(ROOT, 'profiles')


def rd(n):
    return open(os.path.join(E, n)).read()


def put(name, header, body):
    with open(os.path.join(P, name), 'w') as f:
        f.write((header.rstrip('\n') + '\n' if header else '') + body)


for f in os.listdir(E):
    if f.startswith(f'{rn}_pmc_') and (f.endswith('.json') or f.endswith('.csv')):
        shutil.copy(os.path.join(E, f), os.path.join(P, f))
shutil.copy(os.path.join(E, f'{rn}_bench_line.json'), os.path.join(P, f'{rn}_bench_line.json'))
line = json.load(open(os.path.join(P, f'{rn}_bench_line.json')))
sha = line['kernel_source_sha16']
put(f'{rn}_gemm_shapes.txt',
    "per-shape table of the eight Linear products of an EcgVit-base layer (M = 512 x 251 = 128 512 token rows), tools/gemm_ab.py --plain --nt4 --no-old, one process, interleaved rounds\n"
    "'shipped' = what ecgvit_gemm dispatches for the case; '8w' / '4w' = the eight- / four-wave body forced; 'nt' = non-temporal output stores; 'lib plain' = torch.matmul (hipBLASLt), "
    "plain product only: a yardstick, never on the product path", rd(f'{rn}_gemm_shapes.txt'))
put(f'{rn}_gemm_shapes_small.txt',
    "the same table at the EcgVit-small shapes (hidden 512, FFN 2048, M = 256 x 251 = 64 256 token rows: BASELINE.json configs[1]), tools/gemm_ab.py --m 64256 --dim 512 --plain --nt4 --no-old\n"
    "reading: the plain products beat the vendor library by 20-30 % at K = 512 and are level with it at K >= 1536; the two FFN-wide epilogues cost MORE than their products here "
    "(FFN-up forward ~250 us against ~125 plain: 8 K-tiles of main loop per tile against the same 20-k-cycle epilogue)", rd(f'{rn}_gemm_shapes_small.txt'))
put(f'{rn}_attn_ab.txt',
    "attention kernels, round-3 library (csrc/build/libecgvit_hip_r03.so, built from commit 6b13165) against the shipped library, tools/attn_ab.py: one process, one device, interleaved rounds; "
    "512 x 12 x 251 (EcgVit-base), then 512 x 12 x 501 (two key windows); then tools/attn_variants.py (TOOLS build of the shipped sources: stagger / priority variants of the backward, 2 = what ships)\n"
    "(forward: 16-B output stores through v_permlane32_swap, bit-identical; backward: 13.5 -> 9.5 vector instructions per score element, pipelined fragment reads -- same arithmetic with the keep "
    "scale folded into the stored -LSE, hence the last-bit differences in dqkv)", rd(f'{rn}_attn_ab.txt'))
put(f'{rn}_attn_q8_cost.txt',
    "what the 8-bit emitting attention entry points cost over the plain ones (tools/attn_q8_cost.py, one process, interleaved); before this round's fixes the emitting forward took 857 / 597 us "
    "(profiles/r04_amax_atomics.txt)", rd(f'{rn}_attn_q8_cost.txt'))
put(f'{rn}_pmc_attn.txt',
    "counters of the fused attention kernels alone (tools/pmc_attn.sh: tools/attn_only.py under separate rocprofv3 --pmc passes), 512 x 12 x 251, dropout 0.1\n"
    "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; per launch\n" + f"kernel_source_sha16: {sha}", rd(f'{rn}_pmc_attn.txt'))
put(f'{rn}_stress.txt',
    "tools/stress.py 150: randomized exact-integer cases of the streaming GEMM kernels, attention backward (shipped persistent kernel against the one-item kernel of the tools library) on random shapes",
    rd(f'{rn}_stress.txt'))
ab = rd(f'{rn}_step_ab.txt')
vals = {}
for m in re.finditer(r'--hip-lib=(\S+): "value": ([0-9.]+)', ab):
    vals.setdefault(m.group(1), []).append(float(m.group(2)))
(k0, v0), (k1, v1) = list(vals.items())[:2]
a, b = sum(v0) / len(v0), sum(v1) / len(v1)
put(f'{rn}_step_ab.txt',
    "whole-step A/B on ONE device, alternating: round-3 library (built from commit 6b13165) against the shipped library, tools/ab_bench.sh --hip-lib "
    "(bench.py --no-cpu-baseline --no-masked --no-small --no-fp8-large --steps 20 --warmup 5)\n" + f"kernel_source_sha16: {sha} (the shipped library's sources)",
    ab + f"=> {100 * (b / a - 1):+.2f} % ({a:.0f} -> {b:.0f} records/s, {k0} -> {k1}) on this device; the driver-style line of the same call: "
         f"{line['value']:.0f} records/s, {line['ms_per_step']:.2f} ms (profiles/{rn}_bench_line.json)\n" + (note + '\n' if note else ''))
for t in ('base', 'small', 'large_fp8'):
    put(f'{rn}_steady_{t}.txt', '', rd(f'{rn}_steady_{t}.txt'))
print('assembled', rn, 'on sources', sha)
