import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    n = r['Name']
    n = n[:110]
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {n}")
print('total', tot / 1e6)
