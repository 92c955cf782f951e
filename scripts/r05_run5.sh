#!/bin/bash
OUT=gpurun_out/r05e; mkdir -p $OUT
set -x
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_guards.py tests/test_gpu_h16.py tests/test_gpu_round3.py -m gpu -x -q -k "wino or Wino or winograd or step or guard" > $OUT/pytest1.log 2>&1; rc=$?; tail -3 $OUT/pytest1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python3 -m pytest tests/test_gpu_step.py -m gpu -x -q > $OUT/pytest2.log 2>&1; rc=$?; tail -3 $OUT/pytest2.log; [ $rc -eq 0 ] || exit $rc
bash scripts/bench_families.sh 2 old1=.:PP_WINO_GEMM_PERSIST=0,PP_WGRAD_STREAM=0 new1=.:PP_WGRAD_STREAM=0 old2=.:PP_WINO_GEMM_PERSIST=0 new2=. > $OUT/families.log 2>&1
cut -c1-330 $OUT/families.log
