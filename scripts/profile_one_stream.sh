#!/bin/bash
# Per-kernel durations with every kernel ALONE on the chip: the same bench command under rocprofv3 --kernel-trace --stats with the
# weight gradients and the auxiliary path back on the main stream (PP_WGRAD_STREAM=0 PP_AUX_SIDE=0).  The two-stream trace
# (scripts/profile_bench.sh) stretches every launch that shares the chip with the other stream; this one is the table to read a
# kernel's own TB/s from.  Run through gpurun:  scripts/profile_one_stream.sh r06
set -eo pipefail
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_one_stream
mkdir -p "$OUT"
export PP_WGRAD_STREAM=0 PP_AUX_SIDE=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-bn-eval > "$OUT/kt.log" 2>&1
cd "$ROOT"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_one_stream_kernel_stats.csv"
grep -h '^{' "$OUT/kt.log" > "$OUT/${TAG}_one_stream_bench_line_under_rocprof.json" || true
rm -rf "$OUT/kt"
echo "one-stream kernel stats in $OUT"
