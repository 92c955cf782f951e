#!/bin/bash
# Build the kernels of another git revision beside the product library, for same-box A/B runs through PP_LIB_PATH:
#   scripts/build_base.sh <rev> [name]   ->  pacingpseudo_amd/lib/<name>/libpacingpseudo_hip.so   (name defaults to "base")
set -e
REV=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
TMP=/tmp/pp_base_$NAME; rm -rf $TMP; mkdir -p $TMP/csrc $TMP/include
for f in $(git ls-tree --name-only $REV pacingpseudo_amd/csrc/); do git show $REV:$f > $TMP/csrc/$(basename $f); done
for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > $TMP/include/$(basename $f); done
make -j8 CSRC=$TMP/csrc OUT=pacingpseudo_amd/lib/$NAME HIPFLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$TMP/include -I$TMP/csrc -Wno-unused-function -w" all
rm -f pacingpseudo_amd/lib/$NAME/*.o
ls -la pacingpseudo_amd/lib/$NAME/
