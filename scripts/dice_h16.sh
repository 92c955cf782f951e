#!/bin/bash
# Validation-Dice trajectories of the 16-bit storage mode (tests/studies/dice_study.py --storage fp16) for seeds $1..$2,
# pure storage mode and with fp16 operands (--products 1):   scripts/dice_h16.sh 1 12     -> gpurun_out/dice_h16/*.json
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dice_h16
for s in $(seq "$1" "$2"); do
  timeout -k 10 600 python tests/studies/dice_study.py --backend hip --seed $s --storage fp16 --out gpurun_out/dice_h16/r04_hip_h16_s$s.json > gpurun_out/dice_h16/log_h16_s$s.txt 2>&1 || { echo "seed $s failed"; tail -3 gpurun_out/dice_h16/log_h16_s$s.txt; exit 1; }
  timeout -k 10 600 python tests/studies/dice_study.py --backend hip --seed $s --storage fp16 --products 1 --out gpurun_out/dice_h16/r04_hip_h16x1_s$s.json > gpurun_out/dice_h16/log_h16x1_s$s.txt 2>&1 || { echo "seed $s (x1) failed"; exit 1; }
  echo "seed $s done: $(tail -1 gpurun_out/dice_h16/log_h16_s$s.txt | cut -c1-60) | $(tail -1 gpurun_out/dice_h16/log_h16x1_s$s.txt | cut -c1-60)"
done
