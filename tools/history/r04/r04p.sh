cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
for r in 1 2; do
for v in vpl2 ks2a ks2b ks1a ks1b; do
  f=$A/lib_step_k7_$v.so
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py $v GINet 2>/dev/null | grep "^graph" | tee -a $O/ab_ks.txt
done; done
