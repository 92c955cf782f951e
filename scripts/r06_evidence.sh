#!/bin/bash
# Round-6 evidence, one part per gpurun call (a call is limited to 20 minutes):
#   scripts/r06_evidence.sh tests    -> full GPU suite with per-test durations, smoke(), the driver's exact bench command
#   scripts/r06_evidence.sh ab       -> same-box A/B r05 tree (_r05/, built by `git archive <r05 head> | tar -x -C _r05 && make -C _r05`) against
#                                       this tree, and the data-parallel step on ONE rank (PP_FORCE_DIST=1: RCCL communicators, the six
#                                       gradient buckets, the packed loss-denominator all-reduce, the bank broadcast) against the plain step
#   scripts/r06_evidence.sh prof     -> the five rocprofv3 passes of the train-mode-BatchNorm step (profiles/r06_*)
#   scripts/r06_evidence.sh profeval -> the same for the eval-mode-BatchNorm step, the reference's state from epoch 1 on (profiles/r06_evalbn_*)
#   scripts/r06_evidence.sh lease N  -> the driver's exact bench command only (a further lease)
set -o pipefail
PART=${1:-tests}; TAG=r06
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$ROOT"
OUT=gpurun_out/${TAG}_evidence; mkdir -p "$OUT"
need() { [ -e "$1" ] || { echo "r06_evidence: $1 is missing -- $2" >&2; exit 3; }; }
case $PART in
tests)
  rm -f gpurun_out/parity_report.jsonl gpurun_out/branch_choices.jsonl
  timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$OUT/pytest_gpu.log"
  grep -A17 'slowest 15 durations' "$OUT/pytest_gpu.log" > "$OUT/${TAG}_pytest_gpu_tail.log"; tail -1 "$OUT/pytest_gpu.log" >> "$OUT/${TAG}_pytest_gpu_tail.log"
  cp gpurun_out/parity_report.jsonl "$OUT/${TAG}_parity_report.jsonl" 2>/dev/null
  cp gpurun_out/branch_choices.jsonl "$OUT/${TAG}_branch_choices.jsonl" 2>/dev/null
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/${TAG}_smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/${TAG}_smoke.log"
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/leaseA_driver_cmd.json" 2> "$OUT/leaseA.err"; echo "bench rc=$?"; cut -c1-260 "$OUT/leaseA_driver_cmd.json"
  ;;
ab)
  need _r05/pacingpseudo_amd/lib/libpacingpseudo_hip.so "the round-5 baseline tree (see the header of this script)"
  scripts/bench_families.sh 3 r05=_r05 ${TAG}=. > /dev/null 2>&1
  need gpurun_out/bench_families.log "scripts/bench_families.sh wrote no log"
  [ "$(grep -c ' round ' gpurun_out/bench_families.log)" -ge 6 ] || { echo "r06_evidence: the A/B log has fewer than 6 runs" >&2; cat gpurun_out/bench_families.log >&2; exit 3; }
  cp gpurun_out/bench_families.log "$OUT/ab_step_r05_vs_${TAG}.log"; cut -c1-40 "$OUT/ab_step_r05_vs_${TAG}.log"
  scripts/ab_bench.sh 3 plain=. dist1=.:PP_FORCE_DIST=1 > /dev/null 2>&1
  cp gpurun_out/ab_bench.log "$OUT/ab_step_one_rank_rccl_vs_plain.log"; cut -c1-60 "$OUT/ab_step_one_rank_rccl_vs_plain.log"
  ;;
prof)
  scripts/profile_bench.sh "$TAG" > "$OUT/profile.log" 2>&1; echo "profile rc=$?"
  cp gpurun_out/${TAG}_prof/${TAG}_* "$OUT/" 2>/dev/null
  ;;
profeval)
  scripts/profile_bench.sh "${TAG}_evalbn" --bn-mode eval > "$OUT/profile_evalbn.log" 2>&1; echo "profile rc=$?"
  cp gpurun_out/${TAG}_evalbn_prof/${TAG}_evalbn_* "$OUT/" 2>/dev/null
  ;;
lease)
  N=${2:-C}
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/lease${N}_driver_cmd.json" 2> "$OUT/lease${N}.err"; echo "bench rc=$?"; cut -c1-260 "$OUT/lease${N}_driver_cmd.json"
  ;;
esac
ls "$OUT"
