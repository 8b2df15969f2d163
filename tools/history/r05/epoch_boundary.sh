#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_boundary; mkdir -p $O
for mode in "" "--cached"; do
  tag=rebuilt; [ -n "$mode" ] && tag=cached
  timeout 120 python tools/r05/epoch_boundary_probe.py $mode > $O/plain_$tag.txt 2>&1 < /dev/null
  rm -rf $O/tr_$tag
  (cd /tmp && timeout 250 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr_$tag -o run --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05/epoch_boundary_probe.py $mode --epochs 4 > $O/traced_$tag.txt 2>&1 < /dev/null)
  python tools/r05/epoch_boundary_report.py $O/tr_$tag > $O/report_$tag.txt 2>&1
  echo "== $tag"; cat $O/plain_$tag.txt | tail -3; cat $O/report_$tag.txt
  rm -rf $O/tr_$tag
done
