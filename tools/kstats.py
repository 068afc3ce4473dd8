import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2]) if len(sys.argv)>2 else 3
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total ms", round(tot/1e6,2), "per step", round(tot/1e6/n,2))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 40]:
    short=r['Name'].replace('void ','')[:64]
    print(f"{int(r['Calls'])//n:5d}/step {int(r['TotalDurationNs'])/1e6/n:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {short}")
