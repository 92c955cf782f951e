#!/bin/bash
# Quick per-kernel time table of the bench step (rocprofv3 kernel trace, 3 steps):  scripts/ktrace.sh <tag> [ENV=VAL ...]
TAG=${1:-kt}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bn-eval > "$OUT/kt.log" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$OUT/kt"
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import os
# 2 warm-up + 5 timed + 2 event-profiled steps, + 1 + 2 single-stream steps when the second stream is on (bench.py's
# `single_stream` leg; ADVICE r04: the count was stale and per-step figures read 33 % high)
steps = 9.0 + (3.0 if os.environ.get('PP_WGRAD_STREAM', '1') != '0' else 0.0)
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'kernel time per step {tot / steps / 1e6:.2f} ms over {sum(int(r["Calls"]) for r in rows) / steps:.0f} launches')
for r in rows[:32]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:6.2f} ms {int(r['Calls']) / steps:5.1f}x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:90]}")
P
