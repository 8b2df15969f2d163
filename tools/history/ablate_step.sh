#!/bin/bash
# Ablation profiling of the fused step kernel: one library per variant.
#   skipK -> -DDRGNN_SKIP=K (phase K does no work; see PH() in drgnn_step.h)
#   exitK -> -DDRGNN_EXIT_AFTER=K (all workgroups return after barrier K)
# Usage: tools/ablate_step.sh build "<variants>"  (here)  |  tools/ablate_step.sh run "<variants>" [net]  (GPU box)
set -e
cd "$(dirname "$0")/../deeprank-gnn_amd/csrc"
VARIANTS="${2:-base skip1 skip2 skip3 skip4 skip5 skip6 skip7 skip8 skip9 skip10 skip11 skip12 skip13 skip14 skip15 skip16}"
flags() {
  case "$1" in
    base) echo "" ;;
    skip*) echo "-DDRGNN_SKIP=${1#skip}" ;;
    exit*) echo "-DDRGNN_EXIT_AFTER=${1#exit}" ;;
    *) echo "$DRGNN_VARIANT_FLAGS" ;;
  esac
}
if [ "$1" = build ]; then
  mkdir -p ablate
  for v in $VARIANTS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable \
        $(flags $v) -shared -o ablate/libdrgnn_$v.so drgnn_capi.hip 2>&1 | grep -i " error" || true ) &
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 1; done
  done
  wait
  ls ablate | wc -l
else
  cd ../..
  for v in $VARIANTS; do
    DRGNN_LIB=deeprank-gnn_amd/csrc/ablate/libdrgnn_$v.so python tools/time_step.py $v ${3:-GINet}
  done
fi
