#!/bin/bash
# Same-box A/B of whole-step time (GPU box, through gpurun): alternates the variants so clock drift hits all of them alike.
#   scripts/ab_bench.sh <rounds> <label>=<dir>[:ENV=VAL[,ENV=VAL]] ...   -> gpurun_out/ab_bench.log (one bench JSON line per run)
# e.g. scripts/ab_bench.sh 3 r02=_r02 r03=. wino128=.:PP_WINO_MIN_CIN=128
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
ROUNDS=$1; shift
: > gpurun_out/ab_bench.log
for r in $(seq 1 "$ROUNDS"); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; dir=${rest%%:*}; envs=""
    [ "$rest" != "$dir" ] && envs=$(echo "${rest#*:}" | tr ',' ' ')
    line=$(cd "$dir" && env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bn-eval 2>/dev/null | grep '^{' | tail -1)
    ms=$(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], {f["family"][:10]: f["ms_per_step"] for f in d["roofline"].get("matrix_families", [])})' 2>/dev/null)
    echo "$label round $r: $ms" | tee -a gpurun_out/ab_bench.log
  done
done
