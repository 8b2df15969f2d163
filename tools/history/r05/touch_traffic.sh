#!/bin/bash
# How much of the step launches' counter traffic is the builder's L2 touch of the tile rows it is about to write (topo_touch_*)?
# FETCH_SIZE / WRITE_SIZE passes (own --pmc runs, kernel trace only) of bench.py for the shipped library and a -DDRGNN_NO_TOUCH build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_touch; mkdir -p $O
for net in ${NETS:-GINet sGAT FoutNet}; do for lib in base notouch; do
  if [ $lib = base ]; then L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; else L=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_notouch.so; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && DRGNN_LIB=$L timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $O/${net}_${lib}_$ctr -o r --output-format csv -- \
       python $GRAFT_REPO_ROOT/bench.py --net $net --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-other-nets > $O/${net}_${lib}_$ctr.log 2>&1 < /dev/null)
    f=$(find $O/${net}_${lib}_$ctr -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$net" "$lib" "$ctr" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "_co_topo" not in row["Kernel_Name"]:
        continue
    a = acc[row["Kernel_Name"][:60]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-8s %-8s %-10s %-62s %10.1f KB per launch (%d launches)" % (sys.argv[2], sys.argv[3], sys.argv[4], k, v / max(n, 1), n))
PY
    rm -rf $O/${net}_${lib}_$ctr
  done
done; done
