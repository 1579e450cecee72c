"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: launches and per-launch mean of every counter."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
if not f:
    sys.exit("no counter_collection.csv under " + sys.argv[1])
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r.get("Kernel_Name", "")
    k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*$", "", k)[:70]  # drop the argument list, keep template arguments
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r.get("Dispatch_Id"))
names = sorted({c for v in agg.values() for c in v})
key = names[0]
print("kernel".ljust(70), "launches", " ".join(n.rjust(26) for n in names))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k][key])):
    n = max(1, len(disp[k]))
    print(k.ljust(70), str(n).rjust(8), " ".join(f"{agg[k].get(c, 0.0) / n:26.1f}" for c in names))
