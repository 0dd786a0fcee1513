#!/bin/bash
# counters of the fused attention kernels alone (separate --pmc passes, kernel-trace only): bash tools/pmc_attn.sh [N] [p]
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_attn; rm -rf $O; mkdir -p $O
N=${1:-251}; P=${2:-0.1}
python3 $R/tools/attn_only.py 10 $N $P > $O/plain.txt 2>&1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  T=$(echo $C | tr ' ' '+' | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/attn_only.py 4 $N $P > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$O/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = 'bwd' if 'bwd' in r['Kernel_Name'] else ('fwd' if 'attn_fwd' in r['Kernel_Name'] else None)
        if not k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]): print(f'   {c:34s} {agg[k][c] / cnt[(k, c)]:16.0f} per launch')
PY
cat $O/plain.txt
