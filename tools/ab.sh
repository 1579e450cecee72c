#!/bin/bash
# graph-level step time of the default library (A) and the A/B library (B), alternating, on one box:  bash tools/ab.sh [repeats]
N=${1:-3}
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then export VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so; else unset VMM_LIB_PATH; fi
  python bench.py --no-train --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); f=d['denoiser_ms_by_kernel_family']; print('lib=$v', d['ms_per_step'], 'conv3x3', f['vmm_conv3x3_bf16x3'], 'proj', f.get('vmm_proj_bf16x3'), 'tcore', f.get('vmm_temporal_core_bf16x3'))"
done; done
