import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
steps=float(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", tot/1e6/steps, "launches per step", sum(int(r["Calls"]) for r in rows)/steps)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(f'{r["Name"][:100]:100s} calls/step={int(r["Calls"])/steps:6.1f} avg_us={float(r["AverageNs"])/1e3:9.1f} ms/step={float(r["TotalDurationNs"])/1e6/steps:7.3f}')
