#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE (they do not fit one
# pass on gfx950), kernel-trace only (no sys/hip/hsa trace domains).  Output: gpurun_out/pmc_bench/summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bench; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm_bf16_q_kernel' in n: k = 'gemm_q'                          # the dominant symbol: persistent A.B^T GEMM (forward + dgrad)
        elif 'gemm_bf16_pers_kernel<true, true' in n: k = 'gemm_nt'
        elif 'gemm_bf16_v2_kernel<true, true' in n: k = 'gemm_nt_patch'     # patch-embed (K = 240: generic DMA path)
        elif 'gemm_bf16_v2_kernel<true, false' in n: k = 'gemm_nn'
        elif 'gemm_bf16_tq_kernel' in n: k = 'gemm_tq'                        # streaming weight-gradient GEMM
        elif 'gemm_bf16_v2_kernel<false, false' in n: k = 'gemm_tn'
        elif 'attn_fwd' in n: k = 'attn_fwd'
        elif 'attn_bwd' in n: k = 'attn_bwd'
        elif 'layernorm_bwd_fit' in n: k = 'ln_bwd'
        elif 'layernorm_fwd_fit' in n: k = 'ln_fwd'
        elif 'layernorm_bwd' in n: k = 'ln_bwd'
        elif 'layernorm_fwd' in n: k = 'ln_fwd'
        else: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
    o = {c: sum(v) / len(v) for c, v in d.items()}
    o['launches'] = len(next(iter(d.values())))
    if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o:
        # units: KiB; gfx950 correction: FETCH_SIZE reports half of the bytes of wide coalesced streaming reads -> x2
        o['hbm_bytes_per_launch'] = (2.0 * o['FETCH_SIZE'] + o['WRITE_SIZE']) * 1024.0
    out[k] = o
json.dump(out, open('$O/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
