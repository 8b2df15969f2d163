cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
NET=${NET:-FoutNet}; TU=${TU:-6}
for v in base 0 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k${TU}_exit$v.so; fi
  [ -f $f ] || continue
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py exit$v $NET 2>/dev/null | grep "^graph" >> $O/timeline_$NET.txt
done
cat $O/timeline_$NET.txt
