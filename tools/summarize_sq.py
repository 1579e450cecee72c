"""Aggregate a rocprofv3 --pmc pass (SQ_*) per kernel into JSON for profiles/sq_latest.json and print the derived figures:
matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES (pipe-busy cycles summed over every SIMD: 32 per v_mfma_f32_32x32x16_bf16) /
(1024 SIMDs x kernel duration x 2.4 GHz), the duration being the average of the SAME kernel in the un-profiled --kernel-trace --stats
pass of the same command (counter passes serialise and slow the kernels; GRBM_GUI_ACTIVE under the profiler includes the dispatch
overhead) -- i.e. the fraction of the matrix pipes' peak issue capacity at the nominal clock, padded and transposed products included;
share of wave cycles parked at s_waitcnt / barriers = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue stalls = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
(MI355X_MICROARCH.md, PMC slots).

    python tools/summarize_sq.py gpurun_out/prof_<tag>/pmc_sq out.json [kernel_stats.csv]
"""
import csv
import glob
import json
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
    k = re.sub(r"^void ", "", r.get("Kernel_Name", "")).replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*$", "", k)[:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r.get("Dispatch_Id"))

def short(name):
    k = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", k)[:90]


avg_ns = {}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    for r in csv.DictReader(open(sys.argv[3])):
        avg_ns[short(r.get("Name", ""))] = float(r.get("AverageNs", 0) or 0)
table = {}
print(f"{'kernel':70s} {'launches':>8s} {'mfma_util':>9s} {'parked':>7s} {'stall':>7s} {'active':>7s}")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0.0)):
    n = max(1, len(disp[k]))
    row = {c: v / n for c, v in agg[k].items()}
    row["launches"] = n
    wc, gui = row.get("SQ_WAVE_CYCLES", 0.0), row.get("GRBM_GUI_ACTIVE", 0.0)
    if avg_ns.get(k):
        row["avg_ns_unprofiled"] = avg_ns[k]
        row["mfma_pipe_util"] = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * avg_ns[k] * 2.4)
    if wc:
        row["parked_share"] = row.get("SQ_WAIT_ANY", 0.0) / wc
        row["issue_stall_share"] = row.get("SQ_WAIT_INST_ANY", 0.0) / wc
        row["active_share"] = row.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    table[k] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in row.items()}
    print(f"{k[:70]:70s} {n:8d} {row.get('mfma_pipe_util', 0):9.3f} {row.get('parked_share', 0):7.3f} {row.get('issue_stall_share', 0):7.3f} {row.get('active_share', 0):7.3f}")
if len(sys.argv) > 2:
    json.dump(table, open(sys.argv[2], "w"), indent=1, sort_keys=True)
