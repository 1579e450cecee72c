#!/bin/bash
# A/B of an environment setting on the training step (same library, one box):  bash tools/ab_env_train.sh "VMM_S2_SPLIT_BELOW=128" [repeats]
SET="$1"; N=${2:-2}
for i in $(seq $N); do for v in A B; do
  if [ $v = B ]; then PRE="env $SET"; else PRE=""; fi
  $PRE python tools/time_train_modes.py bf16x3 2>/dev/null | grep -v amdgpu | sed "s/^/$v /" | cut -c1-420
done; done
