#!/bin/bash
# A/B of an environment setting on one box (same library), every family of the captured sampling step printed:
#   bash tools/ab_env.sh "VMM_C3_NJ1=1" [repeats] [extra bench flags]
SET="$1"; N=${2:-3}; shift; shift
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then PRE="env $SET"; else PRE=""; fi
  $PRE python bench.py --no-train --no-extras --no-cpu-baseline --no-config4 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); f=d['denoiser_ms_by_kernel_family']
print('$v', d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('vmm_','').replace('_bf16x3',''), v) for k, v in list(f.items())[:10]))"
done; done
