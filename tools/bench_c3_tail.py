import subprocess, sys, os
for n in (42, 44, 56, 57, 58, 71, 72, 85, 86, 88, 99, 100):
    r = subprocess.run([sys.executable, "tools/bench_c3_data.py", str(n), "40"], capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith(" 96x96"):
            us = float(l.split("random:")[-1].split("us")[0])
            tiles = n * 36
            print(f"nimg {n:3d} tiles {tiles:5d} rounds {tiles/512:5.2f}  {us:7.1f} us  {us/tiles*512:6.1f} us per round-equivalent  {us/n:5.2f} us/frame", flush=True)
