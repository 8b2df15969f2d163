cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
for modes in "10" ""; do
for v in base 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k6_exit$v.so; fi
  [ -f $f ] || continue
  DRGNN_LAYOUT_MODES="$modes" DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py exit$v FoutNet 2>/dev/null | grep "^graph" | sed "s/^/modes[$modes] /" >> $O/timeline_FoutNet.txt
done; done
cat $O/timeline_FoutNet.txt
