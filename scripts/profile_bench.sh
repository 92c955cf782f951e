#!/bin/bash
# Reproduces every number in profiles/<tag>_* from the CURRENT binary on the GPU box (run through gpurun):
#   scripts/profile_bench.sh r02            -> profiles-ready files under gpurun_out/<tag>_prof/
#   scripts/profile_bench.sh r06_evalbn --bn-mode eval   (further arguments go to bench.py: the eval-mode-BatchNorm step,
#                                                          the reference's state from epoch 1 on, train_chaos.py:370)
# Three separate rocprofv3 runs of the same bench command (the interpreter directly after `--`: no wrapper hop):
#   1. --kernel-trace --stats  -> <tag>_kernel_stats.csv   (per-kernel calls / total / average duration)
#   2. --pmc FETCH_SIZE        -> \
#   3. --pmc WRITE_SIZE        ->  } <tag>_hbm_traffic_per_launch.json via scripts/pmc_traffic.py (gfx950 corrections)
#   4. --pmc SQ_* (8 counters) -> <tag>_sq_counters_per_kernel.json (matrix-pipe utilisation, stall shares)
#   5. --pmc GRBM_GUI_ACTIVE   -> <tag>_effective_clock_per_kernel.json (the clock the chip holds in each kernel)
# Copy the two summaries into profiles/ and commit them; bench.py reads the JSON by exact kernel name.
set -eo pipefail
TAG=${1:-r02}
shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
BENCH="$ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-bn-eval $EXTRA"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 $BENCH > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d "$OUT/sq" -o sq -- python3 $BENCH > "$OUT/sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/clk" -o clk -- python3 $BENCH > "$OUT/clk.log" 2>&1
cd "$ROOT"
python3 scripts/pmc_clock.py "$OUT/clk" "$OUT/${TAG}_effective_clock_per_kernel.json" > "$OUT/clock.log" 2>&1 || true
python3 scripts/pmc_sq.py "$OUT/sq" "$OUT/${TAG}_sq_counters_per_kernel.json" > "$OUT/sq_summary.log"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
python3 scripts/trace_gaps.py "$OUT/kt" "$OUT/${TAG}_idle_gaps.json" > "$OUT/gaps.log" 2>&1 || true
python3 scripts/pmc_traffic.py "$OUT/fetch" "$OUT/write" "$OUT/${TAG}_hbm_traffic_per_launch.json" > "$OUT/traffic.log"
grep -h '^{' "$OUT/kt.log" > "$OUT/${TAG}_bench_line_under_rocprof.json" || true
# keep only the summaries (the raw traces are tens of MB)
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq" "$OUT/clk"
echo "profile summaries in $OUT"
