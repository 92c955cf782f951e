#!/bin/bash
# HIP trajectories of the Dice study from the CURRENT binary (GPU box, ~13 s each):
#   scripts/dice_hip_seeds.sh <name> "<seeds>" [extra dice_study.py flags]   -> gpurun_out/dice/r03_hip_<name>_s<seed>.json
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
NAME=$1; SEEDS=$2; shift 2
mkdir -p gpurun_out/dice gpurun_out/dice_logs
for s in $SEEDS; do
  timeout -k 10 180 python tests/studies/dice_study.py --backend hip --seed $s --out gpurun_out/dice/r03_hip_${NAME}_s$s.json "$@" \
      > gpurun_out/dice_logs/hip_${NAME}_s$s.log 2>&1 || echo "$NAME s$s FAILED"
done
echo "$NAME done: $(ls gpurun_out/dice/r03_hip_${NAME}_s*.json 2>/dev/null | wc -l) trajectories"
