#!/bin/bash
# what FETCH_SIZE counts for the epilogue's row loads (16 B per lane, four adjacent lanes = one 64-B half line): the x-aux input gradient against the plain
# product of the same shape -- the difference of the raw counter is the 789 MB aux tensor, tallied at 1x or at 1/2 (guides/MI355X_MICROARCH.md calibrates only the
# 128-B streaming pattern).  usage (through gpurun): bash tools/pmc_fetch_calibrate.sh
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fetch_cal; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python3 $R/tools/gemm_ab.py --only "$1" --plain --no-lib --no-old --rounds 1 --iters 2 > $O/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'][20:75]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v) * 1024 / 1e6, 1) for c, v in d.items()}, 'MB raw per launch,', len(next(iter(d.values()))), 'launches')
PY
