#!/bin/bash
OUT=gpurun_out/r05c; mkdir -p $OUT
set -x
timeout -k 10 1500 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_step.py tests/test_gpu_parallel.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; [ $rc -eq 0 ] || exit $rc
bash scripts/bench_families.sh 2 base=.:PP_LIB_PATH=pacingpseudo_amd/lib/base/libpacingpseudo_hip.so,PP_WGRAD_STREAM=0 new=.:PP_WGRAD_STREAM=0 new2s=. > $OUT/families.log 2>&1
cat $OUT/families.log
