#!/bin/bash
# Ablation variants of ONE step translation unit (tools/r04/build_ablate2.sh <tu-kind> "<variants>"; a variant is exitK =
# -DDRGNN_EXIT_AFTER=K, skipK = -DDRGNN_SKIP=K, or a bare number = exitK): compressed objects under
# deeprank-gnn_amd/csrc/ablate2/ (linked on the GPU box by tools/r04/link_ablate2.sh: whole libraries would not fit the push)
set -e
TU=$1; KS=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/deeprank-gnn_amd/csrc
OUT=$C/ablate2; mkdir -p $OUT
OPT=-O3; [ "$TU" = 1 ] || [ "$TU" = 2 ] && OPT=-Os
FLAGS="$OPT -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $EXTRA"
build() {
  v=$1
  case "$v" in
    skip*) d="-DDRGNN_SKIP=${v#skip}"; name=$v ;;
    exit*) d="-DDRGNN_EXIT_AFTER=${v#exit}"; name=$v ;;
    *) d="-DDRGNN_EXIT_AFTER=$v"; name=exit$v ;;
  esac
  /opt/rocm/bin/hipcc $FLAGS $d -DDRGNN_TU_KIND=$TU -c -o /tmp/abl_k${TU}_$name.o $C/drgnn_step_tu.hip 2>/dev/null && xz -3 -c /tmp/abl_k${TU}_$name.o > $OUT/step_k${TU}_$name.o.xz && rm /tmp/abl_k${TU}_$name.o
}
for k in $KS; do build $k & while [ $(jobs -r | wc -l) -ge 7 ]; do sleep 1; done; done; wait
for o in capi step_k0 step_k1 step_k2 step_k3 step_k4 step_k5 step_k6 step_k7 step_k8; do xz -3 -c $C/build/$o.o > $OUT/base_$o.o.xz; done
ls $OUT | wc -l
