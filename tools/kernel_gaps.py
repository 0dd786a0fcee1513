#!/usr/bin/env python3
"""idle time BETWEEN kernels of a rocprofv3 --kernel-trace run (second half of the trace = steady steps): span, busy, idle, and which kernels the gaps follow.
usage (on the GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-probe ... ; python3 tools/kernel_gaps.py /tmp/tr
Round 6: base 0.13 % idle, EcgVit-small 0.41 % -- all of it behind the step's one readback (the same-step non-finite check); the kernels of a step run back to back."""
import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True))[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 40% of kernels (steady steps)
n=len(rows); rows=rows[int(n*0.5):]
busy=sum(e-s for s,e,_ in rows); span=rows[-1][1]-rows[0][0]
gaps=[(rows[i+1][0]-rows[i][1], rows[i][2][:60], rows[i+1][2][:60]) for i in range(len(rows)-1)]
pos=[g for g in gaps if g[0]>0]
print('kernels',len(rows),'span ms',span/1e6,'busy ms',busy/1e6,'idle ms',(span-busy)/1e6,'idle %',100*(span-busy)/span)
import collections
c=collections.Counter()
for g,a,b in pos: c[(a.split('<')[0][-40:],)]+=g
for k,v in c.most_common(12): print(round(v/1e3,1),'us after',k)
print('median gap us', sorted(g[0] for g in pos)[len(pos)//2]/1e3, 'n gaps', len(pos), 'overlaps', sum(1 for g in gaps if g[0]<0))
