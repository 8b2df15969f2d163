# rocprofv3 kernel durations inside the native epoch loop (rebuilt), 4096 resident graphs, batch 64: usage ... [net]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NET=${1:-GINet}
O=$PWD/gpurun_out/r05_epoch; mkdir -p $O
(cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats -d $O/st_$NET -o run --output-format csv -- python $GRAFT_REPO_ROOT/tools/epoch_bench.py --graphs 4096 --epochs 3 --only native-epoch --net $NET > $O/stats_$NET.log 2>&1 < /dev/null)
f=$(find $O/st_$NET -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -6 "$f" | cut -d, -f1-4,6-7
grep -h "^{" $O/stats_$NET.log | cut -c1-220
