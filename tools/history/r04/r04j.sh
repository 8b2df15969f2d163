cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
bash tools/r04/link_ablate2.sh > $O/link.log 2>&1
A=deeprank-gnn_amd/csrc/ablate2
for v in base 1 2 3 4 5 6 8 9 10 11 12 14; do
  if [ $v = base ]; then f=deeprank-gnn_amd/csrc/libdrgnn.so; else f=$A/lib_step_k7_exit$v.so; fi
  [ -f $f ] || continue
  DRGNN_LIB=$PWD/$f timeout 120 python tools/time_graph.py exit$v GINet 2>/dev/null | grep "^graph" >> $O/timeline_GINet_af.txt
done
DRGNN_LAYOUT_MODES="12" timeout 120 python tools/time_graph.py old GINet 2>/dev/null | grep "^graph" >> $O/timeline_GINet_af.txt
cat $O/timeline_GINet_af.txt
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net GINet > $O/bench.json 2>/dev/null; python -c "
import json;d=json.load(open('$O/bench.json'));print(list(d['roofline']['kernels'].keys()))"
