#!/bin/bash
# builds an experiment variant of the library from a copy of the sources: tools/build_variant.sh <name> "<-D switches>"
# -> deeprank-gnn_amd/csrc/variants/libdrgnn_<name>.so (git-ignored; load with DRGNN_LIB=...)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/drgnn_variant_$NAME
rm -rf $W && mkdir -p $W/pkg $W/include
cp $ROOT/include/drgnn.h $W/include/
mkdir -p $W/pkg/csrc && cp $ROOT/deeprank-gnn_amd/csrc/*.h $ROOT/deeprank-gnn_amd/csrc/*.hip $ROOT/deeprank-gnn_amd/csrc/Makefile $W/pkg/csrc/
make -C $W/pkg/csrc -j6 EXTRA="$EXTRA" >/dev/null 2>$W/err.txt || { tail -20 $W/err.txt; exit 1; }
mkdir -p $ROOT/deeprank-gnn_amd/csrc/variants
cp $W/pkg/csrc/libdrgnn.so $ROOT/deeprank-gnn_amd/csrc/variants/libdrgnn_$NAME.so
echo built variants/libdrgnn_$NAME.so
