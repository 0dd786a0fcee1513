#!/bin/bash
# per-STEP kernel table of the steady state: two kernel-trace passes of the same bench workload with 2 and 6 timed steps; the difference of the
# two --stats tables divided by 4 is what one steady-state step launches and costs (the first step's first-use kernels cancel).
#   usage (through gpurun): bash tools/steady_stats.sh TAG [bench args]     -> gpurun_out/steady_<TAG>.txt
: "${GRAFT_REPO_ROOT:?run through gpurun}"
TAG=${1:?usage: steady_stats.sh TAG [bench args]}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/steady_$TAG; rm -rf "$O"; mkdir -p "$O"
for S in 2 6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$S -- python3 $R/bench.py --steps $S --warmup 1 --no-cpu-baseline --no-probe --no-masked --no-bf16-saved --no-small --no-fp8-large "$@" > $O/s$S.log 2>&1
done
python3 - "$O" "$R" "$TAG" "$@" <<'PY' > $R/gpurun_out/steady_$TAG.txt
import csv, glob, sys, os
O, R, tag = sys.argv[1:4]
sys.path.insert(0, R)
import bench
def load(d):
    p = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(p))}
a, b = load(O + '/s2'), load(O + '/s6')
rows = []
for n, (c6, t6) in b.items():
    c2, t2 = a.get(n, (0, 0.0))
    if c6 - c2 > 0:
        rows.append(((t6 - t2) / 4e6, (c6 - c2) / 4, n))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'steady-state step of: bench.py {" ".join(sys.argv[4:])}  (difference of a 6-step and a 2-step kernel trace, per step)')
print('kernel_source_sha16:', bench.kernel_source_hash())
print(f'{tot:9.3f} ms of kernels per step')
for ms, calls, n in rows[:40]:
    print(f'{calls:7.1f} launches {ms:9.3f} ms  {1e3 * ms / calls:9.1f} us each {100 * ms / tot:5.1f} %  {n[:170]}')
PY
tail -45 $R/gpurun_out/steady_$TAG.txt
