#!/bin/bash
# CPU-oracle trajectories of the Dice study on the GPU box's HOST cores (16 per GPU slot): three at a time, 5 threads each
# (about 13 minutes per trajectory: three seeds fit one 20-minute call).
#   scripts/dice_cpu_on_box.sh "31 32 33 34" -> gpurun_out/dice_cpu/r03_cpu_ref_s<seed>.json  (progress: gpurun_out/dice_logs/)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/dice_cpu gpurun_out/dice_logs
echo $1 | tr ' ' '\n' | xargs -P 3 -I{} sh -c \
  "timeout -k 10 1150 python tests/studies/dice_study.py --backend cpu --seed {} --threads 5 --out gpurun_out/dice_cpu/r03_cpu_ref_s{}.json > gpurun_out/dice_logs/cpu_ref_s{}.log 2>&1; tail -1 gpurun_out/dice_logs/cpu_ref_s{}.log | cut -c1-80"
ls gpurun_out/dice_cpu
