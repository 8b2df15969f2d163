cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_pf; mkdir -p $O
(cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $O/t -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05/cached_prefetch_probe.py GINet 8 > $O/log.txt 2>&1 < /dev/null)
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "k_step3" not in row["Kernel_Name"]:
        continue
    key = (row["Grid_Size_X"], row["Workgroup_Size_X"])
    acc[key][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); acc[key][1] += 1
for k, (v, n) in sorted(acc.items()):
    print("grid %s wg %s: %d launches, avg %.2f us" % (k[0], k[1], n, v / n / 1e3))
PY
tail -3 $O/log.txt
