#!/bin/bash
# fabric bytes of the stamped eight-wave kernel without / with the XCD rendezvous per tile round (tools build): bash tools/pmc_rdv.sh "fwd qkv"
ONLY=${1:-fwd qkv}
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rdv; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$T -- python3 $R/tools/gemm_ab.py --only "$ONLY" --rdv --no-lib --no-old --rounds 1 --iters 3 > $O/$T.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/*/*/*counter_collection.csv')):
    rows = [r for r in csv.DictReader(open(f)) if 'gemm_nt_kernel' in r['Kernel_Name']]
    for r in rows:
        agg[r['Kernel_Name'][:70] + ' grid ' + r.get('Grid_Size', '?')][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for k in sorted(agg):
    d = agg[k]
    # stamped instantiation: dispatches alternate (no rdv, rdv) in issue order of the variants: split by order of first appearance
    ids = sorted({i for v in d.values() for i, _ in v})
    print(k, 'dispatches', len(ids))
    for name, v in d.items():
        by = collections.defaultdict(float)
        for i, x in v:
            by[i] += x
        seq = [by[i] for i in ids]
        print('   ', name, ' '.join('%.0f' % (x * (2 * 1024 / 1e6 if name == 'FETCH_SIZE' else 1e-6)) for x in seq[:16]))
PY
