#!/bin/bash
# scripts/pmc_one.sh <tag> <counters...> -- <python script args>: one rocprofv3 --pmc pass, per-kernel sums to gpurun_out/<tag>.txt
TAG=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift; rocprofv3 --pmc "${CTRS[@]}" --output-format csv -d "$OUT" -o p -- python3 "$SCRIPT" "$@" > "$OUT/run.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    n[(k, r['Counter_Name'])] += 1
for k, d in sorted(acc.items()):
    print(k, {c: round(v / max(n[(k, c)], 1)) for c, v in d.items()}, 'launches', max(n[(k, c)] for c in d))
PY
