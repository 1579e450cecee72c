#!/bin/bash
# A/B of the default library against libvmm_hip_ab.so on one box, every kernel family of the captured sampling step printed:
#   bash tools/ab2.sh [repeats] [extra bench flags]
N=${1:-3}; shift
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then export VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so; else unset VMM_LIB_PATH; fi
  python bench.py --no-train --no-extras --no-cpu-baseline --no-config4 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); f=d['denoiser_ms_by_kernel_family']
print('lib=$v', d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('vmm_','').replace('_bf16x3',''), v) for k, v in list(f.items())[:12]))"
done; done
