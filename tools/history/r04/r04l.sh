cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
for lib in libdrgnn_prof.so libdrgnn_prof8.so; do
  PROF_LIB=$lib timeout 300 python tools/r04/topo_phases.py 0 7 > $O/topo_lean_$lib.txt 2>&1
  PROF_LIB=$lib timeout 300 python tools/r04/topo_phases.py 1 7 > $O/topo_lean_w_$lib.txt 2>&1
done
tail -n 40 $O/topo_lean_*lib*.txt
