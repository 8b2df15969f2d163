#!/bin/bash
# Experiment variant of the library in which ONLY some translation units are rebuilt with extra -D switches (the others are
# the objects of the current build): tools/r05/build_af_variant.sh <name> "<-D switches>" <unit> [<unit> ...]
#   (MAKE_ARGS="OPT=-O3": further make variables, e.g. the optimisation level of the rebuilt units)
#   unit: af_1_32 (GINet two workgroups, 32 wide), af_3_32 (sGAT), af_4_32 (FoutNet), af_2_32, capi, ...
# -> deeprank-gnn_amd/csrc/variants/libdrgnn_<name>.so (git-ignored; load with DRGNN_LIB=...)
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/deeprank-gnn_amd/csrc
W=/tmp/drgnn_afv_$NAME
rm -rf $W && mkdir -p $W/pkg/csrc/build $W/include
cp -p $ROOT/include/drgnn.h $W/include/
cp -p $C/*.h $C/*.hip $C/Makefile $W/pkg/csrc/
cp -p $C/build/*.o $W/pkg/csrc/build/
for u in "$@"; do rm -f $W/pkg/csrc/build/$u.o; done
make -C $W/pkg/csrc -j8 EXTRA="$EXTRA" $MAKE_ARGS libdrgnn.so >/dev/null 2>$W/err.txt || { tail -20 $W/err.txt; exit 1; }
mkdir -p $C/variants
cp $W/pkg/csrc/libdrgnn.so $C/variants/libdrgnn_$NAME.so
echo built variants/libdrgnn_$NAME.so
