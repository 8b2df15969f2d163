cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_fullsize.py tests/test_gpu_epoch.py tests/test_gpu_edge_cases.py -q -x 2>&1 | tail -3
timeout 300 python tools/r04/time_topo.py 64 128 256 2>&1 | grep k_topo | tee $O/time_topo.txt
rm -f gpurun_out/sweep/sweep.txt
SWEEP_SECONDS=1 SWEEP_BATCHES="64 128 256" bash tools/batch_sweep.sh 2>&1 | grep rebuilt
