"""Condense rocprofv3 output directories (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


stats = find("stats/**/*kernel_stats.csv")
if stats:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==", stats)
    rows = list(csv.DictReader(open(stats)))
    for r in rows[:25]:
        name = r.get("Name", "")[:90]
        print(f"{name:90s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
else:
    print("no kernel_stats csv found under", out)

for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{tag}/**/*counter_collection.csv")
    if not f:
        print(f"no counter csv for {tag}")
        continue
    agg, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r.get("Kernel_Name", "")[:90]
        agg[k] += float(r.get("Counter_Value", 0))
        cnt[k] += 1
    print(f"== {counter} per launch (KiB as reported; gfx950: double FETCH_SIZE for wide coalesced reads) ==", f)
    for k in sorted(agg, key=agg.get, reverse=True)[:15]:
        print(f"{k:90s} launches={cnt[k]} total={agg[k]:.0f} per_launch={agg[k] / cnt[k]:.1f}")
