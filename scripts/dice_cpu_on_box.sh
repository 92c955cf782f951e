#!/bin/bash
# CPU-oracle trajectories of the Dice study on the GPU box's HOST cores (16 per GPU slot): three at a time, 5 threads each
# (about 15 minutes per trajectory: three seeds fit one 20-minute call).
#   scripts/dice_cpu_on_box.sh "31 32 33" [name] [extra dice_study.py flags]
#       -> gpurun_out/dice_cpu/r03_cpu_<name>_s<seed>.json   (name defaults to "ref"; progress: gpurun_out/dice_logs/)
#   the 256-px configuration (two at a time, 8 threads each: 10 epochs take ~19 min with three at a time):
#       PAR=2 THR=8 scripts/dice_cpu_on_box.sh "2 3" ref256 --size 256 --epochs 10
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SEEDS=$1; NAME=${2:-ref}; shift; shift
mkdir -p gpurun_out/dice_cpu gpurun_out/dice_logs
echo $SEEDS | tr ' ' '\n' | xargs -P ${PAR:-3} -I{} sh -c \
  "timeout -k 10 1150 python tests/studies/dice_study.py --backend cpu --seed {} --threads ${THR:-5} $* --out gpurun_out/dice_cpu/r03_cpu_${NAME}_s{}.json > gpurun_out/dice_logs/cpu_${NAME}_s{}.log 2>&1; tail -1 gpurun_out/dice_logs/cpu_${NAME}_s{}.log | cut -c1-80"
ls gpurun_out/dice_cpu
