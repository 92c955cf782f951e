#!/bin/bash
OUT=gpurun_out/r05b; mkdir -p $OUT
set -x
python3 scripts/bench_spatial.py > $OUT/spatial_new.log 2>&1 || exit 1
PP_LIB_PATH=pacingpseudo_amd/lib/base/libpacingpseudo_hip.so python3 scripts/bench_spatial.py > $OUT/spatial_base.log 2>&1 || exit 1
grep bilinear $OUT/spatial_base.log $OUT/spatial_new.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; [ $rc -eq 0 ] || exit $rc
bash scripts/bench_families.sh 2 base=.:PP_LIB_PATH=pacingpseudo_amd/lib/base/libpacingpseudo_hip.so new=. > $OUT/families.log 2>&1
cat $OUT/families.log
