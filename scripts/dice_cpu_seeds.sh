#!/bin/bash
# CPU-oracle trajectories of the Dice study for more seeds (build container, background):
#   scripts/dice_cpu_seeds.sh "11 12 13 ..." [parallel] [threads]  -> gpurun_out/dice_cpu/r03_cpu_ref_s<seed>.json
cd "$(dirname "$0")/.."
SEEDS=${1:-"11 12 13 14 15 16 17 18 19 20"}
PAR=${2:-2}
THR=${3:-2}
echo $SEEDS | tr ' ' '\n' | xargs -P "$PAR" -I{} sh -c \
  "[ -f gpurun_out/dice_cpu/r03_cpu_ref_s{}.json ] || python tests/studies/dice_study.py --backend cpu --seed {} --threads $THR --out gpurun_out/dice_cpu/r03_cpu_ref_s{}.json > gpurun_out/dice_logs/cpu_ref_s{}.log 2>&1"
