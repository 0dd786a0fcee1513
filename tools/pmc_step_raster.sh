#!/bin/bash
# HBM-side traffic of gemm_nt_kernel INSIDE the train step for column-group sizes of the tile walk (tools build, ECGVIT_NT_G):
# separate FETCH_SIZE / WRITE_SIZE passes.  usage: bash tools/pmc_step_raster.sh "0 6"
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_step_raster; rm -rf $O; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-probe --no-masked --no-bf16-saved --no-small --no-fp8-large --hip-lib $R/ecg-representation-learning_amd/csrc/build/libecgvit_hip_tools.so"
for G in ${1:-0 6}; do
  export ECGVIT_NT_G=$G
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/g$G-$C -- python3 $R/bench.py $ARGS > $O/g$G-$C.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for G in "${1:-0 6}".split():
    tot = {}
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals = []
        for f in glob.glob('$O/g%s-%s/*/*counter_collection.csv' % (G, C)):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                if 'gemm_nt_kernel' in r['Kernel_Name'] and r['Counter_Name'] == C:
                    per[r['Dispatch_Id']] += float(r['Counter_Value'])
            vals += list(per.values())
        tot[C] = sum(vals) / max(1, len(vals))
    # FETCH_SIZE counts 64-B units per 128-B request on gfx950 (x2), both are reported in KiB
    print('G=%s: per gemm_nt launch  fetch %.3f GB  write %.3f GB  total %.3f GB' % (G, tot['FETCH_SIZE'] * 2 * 1024 / 1e9, tot['WRITE_SIZE'] * 1024 / 1e9, (tot['FETCH_SIZE'] * 2 + tot['WRITE_SIZE']) * 1024 / 1e9))
PY
