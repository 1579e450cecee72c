#!/bin/bash
# One rocprofv3 counter pass over the sampling bench, aggregated per kernel (run through gpurun from the repo root):
#   bash tools/pmc_kernels.sh sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
# (--pmc only, no trace domains beside the kernel trace; at most 8 SQ / 4 TCC counters per pass)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $OUT/raw -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-graph > $OUT/run.log 2>&1
python $ROOT/tools/summarize_pmc.py $OUT/raw > $OUT/summary_$TAG.txt 2>&1
head -60 $OUT/summary_$TAG.txt
