#!/bin/bash
# training-step time and its kernel families, default library (A) against libvmm_hip_ab.so (B), alternating on one box:  bash tools/ab_train2.sh [repeats]
N=${1:-2}
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then export VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so; else unset VMM_LIB_PATH; fi
  python bench.py --steps 8 --no-extras --no-cpu-baseline --no-config4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); t=d['training']; f=t['roofline_training']['ms_by_kernel_family']
print('lib=$v sampling', d['ms_per_step'], 'train', t['ms_per_step'], ' '.join('%s=%.2f' % (k.replace('vmm_','').replace('_bf16x3',''), x) for k, x in list(f.items())[:16]))"
done; done
