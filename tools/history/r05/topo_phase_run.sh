cd $GRAFT_REPO_ROOT
for blk in 0 8; do
  PROF_LIB=variants/libdrgnn_prof.so PHASE_BLOCK=$blk timeout 100 python tools/r04/topo_phases.py 0 7 2>&1 | grep -v "amdgpu.ids"
done
