#!/bin/bash
# CPU-oracle trajectories of the Dice study for more seeds (build container, background):
#   scripts/dice_cpu_seeds.sh "11 12 13 ..." [parallel] [threads]  -> gpurun_out/dice_cpu/r03_cpu_<NAME>_s<seed>.json
#   NAME (default ref) and EXTRA (extra dice_study.py flags) come from the environment, e.g. the 256-px configuration:
#   NAME=ref256 EXTRA="--size 256 --epochs 10" scripts/dice_cpu_seeds.sh "4 5 6" 1 2
cd "$(dirname "$0")/.."
SEEDS=${1:-"11 12 13 14 15 16 17 18 19 20"}
PAR=${2:-2}
THR=${3:-2}
NAME=${NAME:-ref}
mkdir -p gpurun_out/dice_cpu gpurun_out/dice_logs
echo $SEEDS | tr ' ' '\n' | xargs -P "$PAR" -I{} sh -c \
  "[ -f gpurun_out/dice_cpu/r03_cpu_${NAME}_s{}.json ] || python tests/studies/dice_study.py --backend cpu --seed {} --threads $THR $EXTRA --out gpurun_out/dice_cpu/r03_cpu_${NAME}_s{}.json > gpurun_out/dice_logs/cpu_${NAME}_s{}.log 2>&1"
