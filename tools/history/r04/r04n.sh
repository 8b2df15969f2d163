cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
timeout 600 python tools/r04/time_topo.py 64 128 256 2>&1 | grep k_topo | tee $O/time_topo.txt
