#!/bin/bash
# Experiment for the "schedule that shares the chip" question: how much would the step gain if its two halves ran as independent
# streams of kernels, each free to fill the other's HBM-bound or matrix-bound stretches?  Upper bound without touching the engine:
# two PROCESSES on the same GPU, batch 16 each, started together, against one process at batch 32 and one at batch 16 alone.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_two_halves
mkdir -p "$OUT"
cd "$ROOT"
B="--no-cpu-baseline --no-bn-eval --prof-timed none --steps 60 --warmup 10"
python3 bench.py $B --batch 32 > "$OUT/one_b32.json" 2> "$OUT/one_b32.err"
python3 bench.py $B --batch 16 > "$OUT/one_b16.json" 2> "$OUT/one_b16.err"
python3 bench.py $B --batch 16 > "$OUT/two_b16_a.json" 2> "$OUT/two_b16_a.err" &
PA=$!
python3 bench.py $B --batch 16 > "$OUT/two_b16_b.json" 2> "$OUT/two_b16_b.err" &
PB=$!
wait $PA; wait $PB
python3 bench.py $B --batch 32 > "$OUT/one_b32_again.json" 2> "$OUT/one_b32_again.err"
python3 - <<'PY'
import json, os
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r06_two_halves')
def line(n):
    for l in open(os.path.join(out, n)):
        if l.startswith('{'):
            return json.loads(l)
for n in ('one_b32', 'one_b16', 'two_b16_a', 'two_b16_b', 'one_b32_again'):
    d = line(n + '.json')
    print(f"{n:14s} {d['value']:9.1f} images/s  {d['ms_per_step']:7.3f} ms/step  batch {d['config']['global_batch']}")
a, b = line('two_b16_a.json'), line('two_b16_b.json')
print(f"two processes together: {a['value'] + b['value']:.1f} images/s (their timed regions overlap only roughly: both ran 70 steps from a common start)")
PY
