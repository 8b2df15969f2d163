cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
for r in 1 2; do
for v in base fence1 fence2 fence3 fence4; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k7_$v.so; fi
  [ -f $f ] || continue
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py $v GINet 2>/dev/null | grep "^graph" | tee -a $O/fence.txt
done; done
