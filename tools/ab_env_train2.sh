#!/bin/bash
# Training-step time with / without a VMM_DISABLE feature, alternating on one box:  bash tools/ab_env_train2.sh <feature> [repeats] [modes...]
F=$1; N=${2:-2}; shift; shift; MODES=${@:-fp16 bf16x3}
for i in $(seq $N); do
  python tools/time_train_modes.py $MODES 2>&1 | grep -E "^(fp16|bf16x3|bf16|fp32)" | sed 's/^/ON  /' | cut -c1-330
  VMM_DISABLE=$F python tools/time_train_modes.py $MODES 2>&1 | grep -E "^(fp16|bf16x3|bf16|fp32)" | sed "s/^/OFF($F) /" | cut -c1-330
done
