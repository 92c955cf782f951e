#!/bin/bash
# r05: the driver's exact bench command + variants of the timed region's instrumentation, one fresh process each
OUT=gpurun_out/${1:-r05a}; mkdir -p $OUT
set -x
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err || exit 1
python3 bench.py --steps 20 --warmup 5 --prof-timed matrix --no-cpu-baseline --no-bn-eval > $OUT/bench_prof_matrix.json 2>> $OUT/err.log || exit 1
python3 bench.py --steps 20 --warmup 5 --prof-timed none --no-cpu-baseline --no-bn-eval > $OUT/bench_prof_none.json 2>> $OUT/err.log || exit 1
PP_WGRAD_STREAM=0 python3 bench.py --steps 20 --warmup 5 --prof-timed none --no-cpu-baseline --no-bn-eval > $OUT/bench_one_stream.json 2>> $OUT/err.log || exit 1
python3 - $OUT <<'P'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k:v for k,v in j['step_ms'].items() if k not in('all','what')}, j['host']['enqueue_ms'], j['host']['lead_ms_min'], (j.get('storage_fp16') or {}).get('fp32_storage_ms_per_step_adjacent'))
    print('   ', j['step_ms']['all'])
P
