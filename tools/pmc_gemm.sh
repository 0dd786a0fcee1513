#!/bin/bash
# PMC passes for one GEMM shape; separate passes as the guide prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$1_$2_$3; mkdir -p $O
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/gemm_one.py $1 $2 $3 4 > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$O/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16' in r['Kernel_Name'] and 'splitk' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k, 'per-launch mean', sum(v) / len(v), 'n', len(v))
PY
