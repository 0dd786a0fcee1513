#!/usr/bin/env python3
"""copy what tools/evidence_round.sh wrote (gpurun_out/evidence_<round>/) into profiles/ under the committed names, each text table behind a
prose header that says what it is and how it was taken (the `kernel_source_sha16:` line each tool printed stays: tools/check_profiles.py holds
it to the round's bench line).  usage: python tools/evidence_assemble.py r05 ["note for the step A/B"]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rn = sys.argv[1] if len(sys.argv) > 1 else 'r05'
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
put(f'{rn}_gemm_shapes_aux8.txt',
    "the FFN-wide launches with the saved GELU' x mask tensor as e4m3 bytes (ECGVIT_EPI_AUX8: what the bf16 engine runs since round 5), tools/gemm_ab.py --only ffn_ --aux8 --no-old --no-lib; "
    "compare fwd ffn_up / dgrad ffn_down with the bf16-tensor rows of " + f"{rn}_gemm_shapes.txt (same process type, same device)", rd(f'{rn}_gemm_shapes_aux8.txt'))
if os.path.exists(os.path.join(E, f'{rn}_gemm_shapes_auxld0.txt')):
    put(f'{rn}_gemm_shapes_auxld0.txt',
        "diagnostic: the same launches with the saved tensor's row pitch set to 0 (every row reads / writes ONE cache-resident row, results meaningless), tools/gemm_ab.py --only ffn_ --aux-ld0: "
        "what the tensor's HBM stream costs the two launches in its bf16 form (the difference to " + f"{rn}_gemm_shapes.txt)", rd(f'{rn}_gemm_shapes_auxld0.txt'))
for name, header in ((f'{rn}_ln_fold_pricing_raw.txt', "raw table behind " + f"{rn}_ln_fold_pricing.txt: tools/gemm_ab.py --rowaffine [--aux8] at the base and small shapes (the LayerNorm-fold pricing bodies of the tools build next to what ships)"),
                     (f'{rn}_attn_fwd_ab.txt', "tools/attn_fwd_ab.py: the attention forward's workgroup forms of ONE tools build in one process (one item per workgroup / streamed 16 waves / streamed 8 waves x 2 workgroups per CU), "
                                               "plain and 8-bit emitting entry points, 512 x 12 x 251 then 256 x 16 x 501, dropout 0.1; outputs compared bit for bit"),
                     (f'{rn}_fp8_nt_stores.txt', "tools/fp8_nt_ab.py: output-store policy of the PLAIN 8-bit A.B^T products at the EcgVit-large shapes (256 x 501 token rows), default against non-temporal, one process, interleaved, median us.  "
                                                 "Shipped rule (gemm_nt.hip): non-temporal for bf16 outputs > 320 MB, or > 240 MB with K <= 1024")):
    if os.path.exists(os.path.join(E, name)):
        put(name, header, rd(name))
put(f'{rn}_stress.txt',
    "tools/stress.py 150: randomized exact-integer cases of the streaming GEMM kernels, attention backward (shipped persistent kernel against the one-item kernel of the tools library) on random shapes",
    rd(f'{rn}_stress.txt'))
ab = rd(f'{rn}_step_ab.txt')
vals = {}
prev = 'r05' if re.search(r'^r05 base', ab, re.M) else 'r04'
for m in re.finditer(r'^(r04|r05|new) base ([0-9.]+) ([0-9.]+) gemm_us ([0-9.]+) masked ([0-9.]+) small ([0-9.]+) fp8 ([0-9.]+) large_bf16 ([0-9.]+)', ab, re.M):
    vals.setdefault(m.group(1), []).append([float(x) for x in m.groups()[1:]])
mean = {k: [sum(c) / len(c) for c in zip(*v)] for k, v in vals.items()}
names = ['base records/s', 'base ms/step', 'A.B^T us/launch', 'masked', 'small', 'fp8-large', 'large bf16']
summ = '; '.join(f'{n}: {mean[prev][i]:.1f} -> {mean["new"][i]:.1f} ({100 * (mean["new"][i] / mean[prev][i] - 1):+.2f} %)' for i, n in enumerate(names))
put(f'{rn}_step_ab.txt',
    ("whole-LINE A/B on ONE device, alternating: round-5 library (csrc/build/libecgvit_hip_r05.so = csrc of commit b50449c, sha256 eff98bfd...) against the shipped library, tools/ab_r05.sh 3 "
     "(bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-saved, the default line with its nested masked / small / fp8-large runs)\n" if prev == 'r05' else
     "whole-LINE A/B on ONE device, alternating: round-4 library (csrc/build/libecgvit_hip_r04.so = the library of commit 9af6816, run with --bf16-aux: it does not know ECGVIT_EPI_AUX8) against the "
     "shipped library, tools/ab_r04.sh 3 (bench.py --steps 10 --warmup 3 --no-cpu-baseline, the default line with its nested masked / small / fp8-large runs)\n")
    + f"kernel_source_sha16: {sha} (the shipped library's sources)",
    ab + "=> " + summ + f"\nthe driver-style line of the same call: {line['value']:.0f} records/s, {line['ms_per_step']:.2f} ms (profiles/{rn}_bench_line.json)\n" + (note + '\n' if note else ''))
if os.path.exists(os.path.join(E, f'{rn}_attn_ab.txt')):
    put(f'{rn}_attn_ab.txt',
        "the fused attention kernels of the PREVIOUS round's library against the shipped ones, tools/attn_ab.py (one process, seven interleaved rounds, median / min us; outputs compared bit for bit): "
        "512 records x 12 heads x 251 tokens, then 256 x 16 x 501 (two key windows); dropout 0.1.  Round 6: the forward above 256 tokens runs the streamed persistent kernel; nothing else changed",
        rd(f'{rn}_attn_ab.txt'))
for t in ('base', 'small', 'large_fp8'):
    put(f'{rn}_steady_{t}.txt', '', rd(f'{rn}_steady_{t}.txt'))
print('assembled', rn, 'on sources', sha)
