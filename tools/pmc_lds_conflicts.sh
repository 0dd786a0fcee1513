#!/bin/bash
# where do gemm_nt_kernel's LDS bank conflicts come from?  SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS / SQ_INSTS_LDS per launch type (tools/gemm_ab.py,
# one launch type per pass): a long-K plain product (fragment reads dominate, 64 ds_bpermute per 36-K-tile tile) against the epilogue-heavy launches.
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lds; rm -rf $O; mkdir -p $O
for CASE in "dgrad qkv" "fwd qkv" "fwd ffn_up" "dgrad ffn_down"; do
  T=$(echo $CASE | tr ' ' '_')
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/gemm_ab.py --only "$CASE" --no-lib --no-old --rounds 1 --iters 2 > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$O/*/')):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'gemm_nt_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if m:
        print(d.split('/')[-2], {k: round(v) for k, v in m.items()}, 'conflict/active = %.3f' % (m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_ACTIVE_INST_LDS', 1))))
PY
