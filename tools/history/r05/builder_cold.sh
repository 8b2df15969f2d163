# the co-launched builder alone (step workgroups return at once: -DDRGNN_EXIT_AFTER=0): replayed mini-batch vs inside the epoch loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_exit0.so
O=$PWD/gpurun_out/r05_epoch; mkdir -p $O
DRGNN_LIB=$V timeout 120 python tools/time_graph.py exit0 GINet 2>/dev/null | grep "^graph"
(cd /tmp && DRGNN_LIB=$V timeout 250 rocprofv3 --kernel-trace --stats -d $O/st_exit0 -o run --output-format csv -- python $GRAFT_REPO_ROOT/tools/epoch_bench.py --graphs 4096 --epochs 3 --only native-epoch --net GINet > $O/stats_exit0.log 2>&1 < /dev/null)
f=$(find $O/st_exit0 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -4 "$f" | cut -d, -f1-4,6-7
