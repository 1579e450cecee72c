#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r02_a [stats|traffic|sq|all]
# writes gpurun_out/prof_<tag>/... and copies the compact summaries to profiles/<tag>_* on the box's copy of the repo AND to
# gpurun_out/profiles_<tag>/ (which travels back).  Counter passes use --pmc alone (no trace domains besides the kernel trace).
set -u
TAG=${1:-r1}
WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
KEEP=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras"
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
  cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $KEEP/${TAG}_kernel_stats.csv 2>/dev/null
  grep -m1 "^{\"metric\"" $OUT/stats.log > $KEEP/${TAG}_bench.json
fi
# counter passes on the sampling leg only (eager launches), so that the per-kernel figures are those of the launches the roofline object describes;
# a kernel-trace pass of exactly that command gives the un-profiled duration of the same launch mix (the full command above also runs
# the batch-4 training legs through the same template instances)
PMC_CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph --no-train"
if [ "$WHAT" = all ] || [ "$WHAT" = sq ] || [ "$WHAT" = stats ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sampling -o bench -- $PMC_CMD > $OUT/stats_sampling.log 2>&1
  cp $(find $OUT/stats_sampling -name '*kernel_stats.csv' | head -1) $KEEP/${TAG}_kernel_stats_sampling.csv 2>/dev/null
fi
if [ "$WHAT" = all ] || [ "$WHAT" = traffic ]; then
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $PMC_CMD > $OUT/pmc_write.log 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sq ]; then
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d $OUT/pmc_sq -o bench -- $PMC_CMD > $OUT/pmc_sq.log 2>&1
  python $ROOT/tools/summarize_sq.py $OUT/pmc_sq $KEEP/${TAG}_sq.json $KEEP/${TAG}_kernel_stats_sampling.csv > $KEEP/${TAG}_sq_counters.txt 2>&1
  head -30 $KEEP/${TAG}_sq_counters.txt
fi
python $ROOT/tools/summarize_prof.py $OUT $KEEP/${TAG}_traffic.json > $KEEP/${TAG}_rocprofv3_summary.txt 2>&1
tail -45 $KEEP/${TAG}_rocprofv3_summary.txt
