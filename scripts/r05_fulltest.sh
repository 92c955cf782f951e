#!/bin/bash
OUT=gpurun_out/${1:-r05f}; mkdir -p $OUT
shift
timeout -k 10 1700 python3 -m pytest tests -m gpu -x -q "$@" > $OUT/pytest_gpu.log 2>&1; rc=$?; tail -4 $OUT/pytest_gpu.log; exit $rc
