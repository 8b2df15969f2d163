cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
for v in base skip1 skip2 skip3 skip4 skip5 skip6 skip8 skip9 skip10 skip11 skip12 skip13 skip16; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k7_$v.so; fi
  [ -f $f ] || continue
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py $v GINet 2>/dev/null | grep "^graph" >> $O/skip_GINet.txt
done
for v in base skip1 skip2 skip3 skip4 skip5 skip6 skip7 skip8 skip9 skip10 skip11 skip12 skip13 skip16; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k6_$v.so; fi
  [ -f $f ] || continue
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py $v FoutNet 2>/dev/null | grep "^graph" >> $O/skip_FoutNet.txt
done
cat $O/skip_GINet.txt $O/skip_FoutNet.txt
