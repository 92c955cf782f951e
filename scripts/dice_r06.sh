#!/bin/bash
# Validation-Dice trajectories of the ROUND-6 binary: seeds $1..$2 of the 128-px / 40-epoch study, one run each of
# the fp32-grade path, `--storage fp16` and `--storage bf16` (~13 s per run on the MI355X box).  The CPU-oracle side is the committed
# one (profiles/r03_dice_parity.json rows, seeds 11..60):
#   scripts/dice_r06.sh 11 30                        -> gpurun_out/dice_r06/r06_hip_{final,fp16,bf16}_s<seed>.json   (GPU box)
#   python tests/studies/dice_summary.py --from_compact profiles/r03_dice_parity.json --hip gpurun_out/dice_r06 --min_hip_round 6 \
#          --out profiles/r06_dice_parity.json                                                                      (here)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dice_r06
for s in $(seq "$1" "$2"); do
  for v in final fp16 bf16; do
    [ -s gpurun_out/dice_r06/r06_hip_${v}_s$s.json ] && continue
    extra=""; [ $v != final ] && extra="--storage $v"
    timeout -k 10 300 python tests/studies/dice_study.py --backend hip --seed $s $extra --out gpurun_out/dice_r06/r06_hip_${v}_s$s.json \
        > gpurun_out/dice_r06/log_${v}_s$s.txt 2>&1 || { echo "seed $s $v failed"; tail -3 gpurun_out/dice_r06/log_${v}_s$s.txt; exit 1; }
  done
  echo "seed $s done: $(tail -1 gpurun_out/dice_r06/log_final_s$s.txt | cut -c1-60)"
done
