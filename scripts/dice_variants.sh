#!/bin/bash
# HIP trajectories of the Dice study under kernel-selection knobs that change the summation order (not the arithmetic):
#   scripts/dice_variants.sh   -> gpurun_out/dice/r02_hip_<variant>_s<seed>.json   (13 s each on an MI355X)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/dice gpurun_out/dice_logs
run() {   # name, env assignment
  for s in 1 2 3 4 5 6 7 8 9 10; do
    env $2 timeout -k 10 120 python tests/studies/dice_study.py --backend hip --seed $s --out gpurun_out/dice/r02_hip_$1_s$s.json \
        > gpurun_out/dice_logs/hip_$1_s$s.log 2>&1 || echo "$1 s$s FAILED"
  done
  echo "$1 done"
}
run unfusedbn PP_FUSE_BN=0
run wgradmp0 PP_WGRAD_MP=0
run wino128 PP_WINO_MIN_CIN=128
run nohalo PP_CONV_HALO_F16=0
run wgradfp32 PP_WGRAD_H16_OFF=1
run maxn96 PP_HALO_F16_MAXN=96
