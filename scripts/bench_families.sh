#!/bin/bash
# Same-box A/B with the per-family time table (HBM-bound families included): like scripts/ab_bench.sh, but prints every
# family of the `kernels` block of the bench line.   scripts/bench_families.sh <rounds> <label>=<dir>[:ENV=VAL,...] ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
ROUNDS=$1; shift
: > gpurun_out/bench_families.log
for r in $(seq 1 "$ROUNDS"); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; dir=${rest%%:*}; envs=""
    [ "$rest" != "$dir" ] && envs=$(echo "${rest#*:}" | tr ',' ' ')
    line=$(cd "$dir" && env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bn-eval 2>/dev/null | grep '^{' | tail -1)
    echo "$line" | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
k = d.get("kernels", {})
fam = {n: round(v["ms_per_step"], 3) if isinstance(v, dict) else v for n, v in k.items()} if isinstance(k, dict) else k
one = (d.get("roofline", {}).get("single_stream") or {}).get("all_families_ms_per_step")
print(sys.argv[1], "round", sys.argv[2], d["ms_per_step"], fam, "| alone on the chip:", one)' "$label" "$r" | tee -a gpurun_out/bench_families.log
  done
done
