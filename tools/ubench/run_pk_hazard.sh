#!/bin/bash
# tools/ubench/pk_hazard under multi-process load: P concurrent processes per sequence, then every sequence alone.   bash tools/ubench/run_pk_hazard.sh [P=4] [launches=300]

cd "$(dirname "$0")/../.."
P=${1:-4}; L=${2:-300}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/pk_hazard.hip -o tools/ubench/pk_hazard || exit 2
for seq in 0 1 2 3 4; do
  echo "== seq $seq, $P processes"
  for p in $(seq $P); do tools/ubench/pk_hazard $seq $L & done; wait
done
echo "== mixed: seq 0 next to three other sequences"
tools/ubench/pk_hazard 0 $L & tools/ubench/pk_hazard 1 $L & tools/ubench/pk_hazard 2 $L & tools/ubench/pk_hazard 3 $L & wait
echo "== alone"
for seq in 0 1 2 3 4; do tools/ubench/pk_hazard $seq $L; done
