#!/bin/bash
# Everything the round's evidence files are made from, from the CURRENT binary, in one call on the GPU box:
#   scripts/final_evidence.sh r04     -> gpurun_out/<tag>_evidence/ (copy the summaries into profiles/)
# 1. the full GPU test suite (parity report + branch-choice counts are written by the tests)   2. __graft_entry__.smoke()
# 3. bench.py with its CPU baseline (the line the docs quote)   4. same-box A/B against the previous round's tree (_r0N, if present)
# 5. scripts/profile_bench.sh (rocprofv3 kernel stats, HBM traffic, SQ counters, clocks)
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_evidence
mkdir -p "$OUT"
rm -f gpurun_out/parity_report.jsonl gpurun_out/branch_choices.jsonl
timeout -k 10 1150 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.jsonl "$OUT/${TAG}_parity_report.jsonl" 2>/dev/null
cp gpurun_out/branch_choices.jsonl "$OUT/${TAG}_branch_choices.jsonl" 2>/dev/null
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 10 600 python bench.py > "$OUT/bench.log" 2>&1; echo "bench rc=$?"
grep -h '^{' "$OUT/bench.log" | tail -1 > "$OUT/${TAG}_bench_line.json"; cut -c1-200 "$OUT/${TAG}_bench_line.json"
PREV=$(ls -d _r0[0-9] 2>/dev/null | tail -1)      # the previous round's tree, if a copy is there (same-box A/B)
if [ -n "$PREV" ]; then scripts/bench_families.sh 3 ${PREV#_}=$PREV ${TAG}=. > /dev/null 2>&1; cp gpurun_out/bench_families.log "$OUT/ab_step_${PREV#_}_vs_${TAG}.log"; cut -c1-60 "$OUT/ab_step_${PREV#_}_vs_${TAG}.log"; fi
scripts/profile_bench.sh "$TAG" > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
cp gpurun_out/${TAG}_prof/${TAG}_* "$OUT/" 2>/dev/null
ls "$OUT"
