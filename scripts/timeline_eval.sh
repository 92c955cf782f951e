#!/bin/bash
# Two-stream timeline of one EVAL-mode-BatchNorm step (the reference's state from epoch 1 on):  scripts/timeline_eval.sh <tag> [step index]
TAG=${1:-tle}; WHICH=${2:-12}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/kt.log" 2>&1
python3 $ROOT/scripts/timeline.py "$OUT/kt" "$OUT/timeline.tsv" $WHICH
rm -rf "$OUT/kt"
