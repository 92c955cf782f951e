#!/bin/bash
# The 256-px / 10-epoch variant of scripts/dice_h16.sh (the benchmark geometry; CPU rows: r03 `ref256`, seeds 1..8): the 16-bit
# storage mode, the same with fp16 operands, and the fp32-storage path of the SAME binary for a paired comparison on any seed.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dice_h16_256
for s in $(seq "$1" "$2"); do
  for v in h16 h16x1 f32; do
    [ -s gpurun_out/dice_h16_256/r04_hip_${v}_s$s.json ] && continue
    extra="--storage fp16"; [ $v = h16x1 ] && extra="--storage fp16 --products 1"; [ $v = f32 ] && extra=""
    timeout -k 10 900 python tests/studies/dice_study.py --backend hip --seed $s --size 256 --epochs 10 $extra --out gpurun_out/dice_h16_256/r04_hip_${v}_s$s.json > gpurun_out/dice_h16_256/log_${v}_s$s.txt 2>&1 || { echo "seed $s $v failed"; tail -3 gpurun_out/dice_h16_256/log_${v}_s$s.txt; exit 1; }
  done
  echo "seed $s done: $(tail -1 gpurun_out/dice_h16_256/log_h16_s$s.txt | cut -c1-50)"
done
