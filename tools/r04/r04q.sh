cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiles or lean" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
timeout 600 python tools/r04/time_topo.py 64 128 256 2>&1 | grep k_topo | tee $O/time_topo.txt
