#!/bin/bash
OUT=gpurun_out/r05e; mkdir -p $OUT
bash scripts/bench_families.sh 2 c256=. c192=.:PP_WINO_MIN_CIN=192 c128=.:PP_WINO_MIN_CIN=128 > $OUT/families_wino_min.log 2>&1
cut -c1-330 $OUT/families_wino_min.log
