#!/bin/bash
# Product library (A) against the A/B library (B = videometamaterials_amd/libvmm_hip_ab.so, tools/build_ab.py), alternating on one box:
#   bash tools/ab_lib.sh [repeats] -- what: fp16 / bf16x3 training step, configs[3] bf16 forward, guided sampling step
N=${1:-2}
AB=$PWD/videometamaterials_amd/libvmm_hip_ab.so
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then export VMM_LIB_PATH=$AB; else unset VMM_LIB_PATH; fi
  python tools/time_train_modes.py fp16 2>&1 | grep -E "^fp16" | cut -c1-200 | sed "s/^/lib=$v train /"
  python tools/bench_hires.py 8 bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$v configs[3] bf16 forward', d['denoiser_forward_ms'], 'conv3x3', d['ms_by_family'].get('vmm_conv3x3_bf16'))"
done; done
