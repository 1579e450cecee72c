#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r1
# writes gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/... and a compact summary gpurun_out/prof_<tag>/summary_<tag>.txt
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
# counter passes on the sampling leg only, so that the per-kernel traffic is that of the launches the roofline object describes
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD --no-graph --no-train > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $CMD --no-graph --no-train > $OUT/pmc_write.log 2>&1
python $ROOT/tools/summarize_prof.py $OUT $OUT/traffic_$TAG.json > $OUT/summary_$TAG.txt 2>&1
tail -40 $OUT/summary_$TAG.txt
