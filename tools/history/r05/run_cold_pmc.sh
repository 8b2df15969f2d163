# warm vs cold step launches under TLB / cache counters (one --pmc pass per counter group; --kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_pmc; mkdir -p $O
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  for which in warm cold; do
    (cd /tmp && COLD_ONLY=$which timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/${which}_$i -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05/cold_timeline.py pmc GINet 32 0 > $O/${which}_$i.log 2>&1 < /dev/null)
    f=$(find $O/${which}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" "$which" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "k_step3" not in row["Kernel_Name"]:
        continue
    a = acc[row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-5s %-40s %14.1f per launch (%d launches)" % (sys.argv[2], k, v / max(n, 1), n))
PY
  done
done
