#!/bin/bash
# training-step time of the default library (A) and the A/B library (B), alternating, on one box:  bash tools/ab_train.sh [repeats]
N=${1:-3}
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then export VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so; else unset VMM_LIB_PATH; fi
  python bench.py --steps 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); t=d['training']; print('lib=$v sampling', d['ms_per_step'], 'train fp32', t['ms_per_step'], 'bf16x3', t['split_bf16_variant']['ms_per_step'])"
done; done
