#!/bin/bash
# on the GPU box: link the variants built by build_ablate2.sh -> deeprank-gnn_amd/csrc/ablate2/lib_step_k<tu>_<variant>.so
C=deeprank-gnn_amd/csrc; A=$C/ablate2; W=/tmp/abl; mkdir -p $W
for f in $A/base_*.o.xz; do b=$(basename $f .o.xz); xz -dc $f > $W/$b.o; done
for f in $A/step_k*_*.o.xz; do
  b=$(basename $f .o.xz); tu=$(echo $b | sed 's/step_k\([0-9]\)_.*/\1/'); xz -dc $f > $W/$b.o
  objs=""; for o in capi step_k0 step_k1 step_k2 step_k3 step_k4 step_k5 step_k6 step_k7 step_k8; do if [ "$o" = "step_k$tu" ]; then objs="$objs $W/$b.o"; else objs="$objs $W/base_$o.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $A/lib_${b}.so $objs &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done; wait
ls $A/*.so | wc -l
