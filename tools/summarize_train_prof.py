import csv,glob,re,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
n=int(sys.argv[2])
tot=0
for r in csv.DictReader(open(f)):
    name=re.sub(r"^void ","",r["Name"]).replace("(anonymous namespace)::","")
    name=re.sub(r"\(.*$","",name)[:80]
    t=float(r["TotalDurationNs"])/n/1e6; tot+=t
    if t>0.3: print(f"{name:80s} calls/step={int(r['Calls'])/n:7.1f} ms/step={t:8.3f}")
print("total",tot)
