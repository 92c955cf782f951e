#!/bin/bash
# Two-stream timeline of one bench step:  scripts/timeline.sh <tag> [ENV=VAL ...]   -> gpurun_out/<tag>/timeline.tsv
TAG=${1:-tl}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bn-eval > "$OUT/kt.log" 2>&1
python3 $ROOT/scripts/timeline.py "$OUT/kt" "$OUT/timeline.tsv" 3
rm -rf "$OUT/kt"
