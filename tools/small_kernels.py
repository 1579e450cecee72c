"""Durations of the small / latency-bound kernels of a training-step trace (rocprofv3 --kernel-trace of tools/profile_train.py): per kernel name the
number of launches per step, the milliseconds per step, the median and the largest launch.   python tools/small_kernels.py <trace dir> <steps>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2])
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
for k, v in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    v2 = sorted(v)
    print(f"{k:50s} launches/step {len(v) / n:7.1f}  ms/step {sum(v) / n / 1e3:7.3f}  median {v2[len(v2) // 2]:8.1f} us  max {v2[-1]:8.1f} us")
