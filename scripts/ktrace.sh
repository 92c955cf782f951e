#!/bin/bash
# quick per-dispatch trace: scripts/ktrace.sh <tag>
set -eo pipefail
TAG=${1:-kt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_kt
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bn-eval > "$OUT/kt.log" 2>&1
cd "$ROOT"
python3 scripts/trace_gaps.py "$OUT/kt" "$OUT/${TAG}_idle_gaps.json" > "$OUT/gaps.log" 2>&1 || true
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$OUT/kt"
