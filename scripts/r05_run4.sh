#!/bin/bash
OUT=gpurun_out/r05d; mkdir -p $OUT
bash scripts/bench_families.sh 2 two=. after=.:PP_WGRAD_AFTER_DGRAD=1 after256=.:PP_WGRAD_AFTER_DGRAD=1,PP_WGRAD_CUS_SIDE=256 after224=.:PP_WGRAD_AFTER_DGRAD=1,PP_WGRAD_CUS_SIDE=224 after160=.:PP_WGRAD_AFTER_DGRAD=1,PP_WGRAD_CUS_SIDE=160 > $OUT/families_after.log 2>&1
cut -c1-330 $OUT/families_after.log
