"""Condense rocprofv3 output directories (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into a small text summary and a
per-kernel traffic table (JSON) for profiles/.

    python tools/summarize_prof.py gpurun_out/prof_<tag> [traffic.json]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)[:90]


stats = find("stats/**/*kernel_stats.csv")
if stats:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(stats)))
    for r in rows[:40]:
        print(f"{short(r.get('Name', '')):74s} calls={r.get('Calls'):>6s} total_ns={r.get('TotalDurationNs'):>12s} avg_ns={float(r.get('AverageNs')):>12.0f} pct={r.get('Percentage')}")
else:
    print("no kernel_stats csv found under", out)

traffic = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{tag}/**/*counter_collection.csv")
    if not f:
        print(f"no counter csv for {tag}")
        continue
    agg, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r.get("Kernel_Name", ""))
        agg[k] += float(r.get("Counter_Value", 0))
        disp[k].add(r.get("Dispatch_Id"))
    print(f"== {counter} per launch, KiB as reported (gfx950: FETCH_SIZE counts 128-B requests at 64 B -> double it for wide coalesced reads) ==")
    for k in sorted(agg, key=agg.get, reverse=True)[:25]:
        n = max(1, len(disp[k]))
        print(f"{k:74s} launches={n:6d} per_launch_KiB={agg[k] / n:12.1f}")
        traffic.setdefault(k, {})[counter + "_KiB_per_launch"] = round(agg[k] / n, 1)
        traffic[k]["launches_" + counter] = n
if len(sys.argv) > 2:
    json.dump(traffic, open(sys.argv[2], "w"), indent=1, sort_keys=True)
