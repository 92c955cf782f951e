#!/bin/bash
# Round-5 evidence in two gpurun calls (a call is limited to 20 minutes):
#   scripts/r05_evidence.sh tests   -> full GPU suite, smoke(), the driver's exact bench command (lease A)
#   scripts/r05_evidence.sh prof    -> the driver's exact bench command (lease B), same-box A/B r04 vs r05, five rocprofv3 passes
#   scripts/r05_evidence.sh lease   -> the driver's exact bench command only (a further lease)
set -o pipefail
PART=${1:-tests}; TAG=r05
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
OUT=gpurun_out/${TAG}_evidence; mkdir -p "$OUT"
case $PART in
tests)
  rm -f gpurun_out/parity_report.jsonl gpurun_out/branch_choices.jsonl
  timeout -k 10 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$OUT/pytest_gpu.log"
  cp gpurun_out/parity_report.jsonl "$OUT/${TAG}_parity_report.jsonl" 2>/dev/null
  cp gpurun_out/branch_choices.jsonl "$OUT/${TAG}_branch_choices.jsonl" 2>/dev/null
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/leaseA_driver_cmd.json" 2> "$OUT/leaseA.err"; echo "bench rc=$?"; cut -c1-260 "$OUT/leaseA_driver_cmd.json"
  ;;
prof)
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/leaseB_driver_cmd.json" 2> "$OUT/leaseB.err"; echo "bench rc=$?"; cut -c1-260 "$OUT/leaseB_driver_cmd.json"
  scripts/bench_families.sh 3 r04=_r04 ${TAG}=. > /dev/null 2>&1; cp gpurun_out/bench_families.log "$OUT/ab_step_r04_vs_${TAG}.log"; cut -c1-40 "$OUT/ab_step_r04_vs_${TAG}.log"
  scripts/profile_bench.sh "$TAG" > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
  cp gpurun_out/${TAG}_prof/${TAG}_* "$OUT/" 2>/dev/null
  ;;
lease)
  N=${2:-C}
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/lease${N}_driver_cmd.json" 2> "$OUT/lease${N}.err"; echo "bench rc=$?"; cut -c1-260 "$OUT/lease${N}_driver_cmd.json"
  ;;
esac
ls "$OUT"
