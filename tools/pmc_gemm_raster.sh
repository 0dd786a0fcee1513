#!/bin/bash
# fabric traffic and L2 hit rate of gemm_nt_kernel on one Linear shape for several column-group sizes (tools build)
# usage: bash tools/pmc_gemm_raster.sh "fwd qkv" "0,1,3,5"
ONLY=${1:-fwd qkv}; GROUPS_=${2:-0,3}
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_raster; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/gemm_ab.py --only "$ONLY" --groups $GROUPS_ --no-lib --no-old --rounds 1 --iters 3 > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/*/*/*counter_collection.csv')):
    rows = [r for r in csv.DictReader(open(f)) if 'gemm_nt_kernel' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    for r in rows:
        agg[ids.index(int(r['Dispatch_Id']))][r['Counter_Name']].append(float(r['Counter_Value']))
for i in sorted(agg):
    d = {k: sum(v) / len(v) for k, v in agg[i].items()}
    print('dispatch', i, 'fetch MB %.0f' % (d.get('FETCH_SIZE', 0) * 2 * 1024 / 1e6), 'L2 hit %.3f' % (d.get('TCC_HIT_sum', 0) / max(1, d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0))))
PY
