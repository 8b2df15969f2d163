cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_prof; mkdir -p $O
for which in warm cold; do
  (cd /tmp && COLD_ONLY=$which timeout 150 rocprofv3 --kernel-trace --stats -d $O/$which -o $which --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05/cold_timeline.py prof GINet 32 1 > $O/$which.log 2>&1 < /dev/null)
  f=$(find $O/$which -name "*kernel_stats.csv" | head -1)
  echo "== $which $f"
  [ -n "$f" ] && head -5 "$f" | cut -c1-200
done
