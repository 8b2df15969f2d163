# SQ LDS counters of the GINet step kernel truncated after every barrier (-DDRGNN_EXIT_AFTER=k builds): cumulative per launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_lds; mkdir -p $O
V=$PWD/deeprank-gnn_amd/csrc/variants
for v in exit1 exit3 exit4 exit5 exit6 exit8 exit9 exit10 exit11 exit12 exit14 full; do
  if [ $v = full ]; then L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; else L=$V/libdrgnn_$v.so; fi
  i=0
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
    i=$((i+1))
    (cd /tmp && DRGNN_LIB=$L timeout 100 rocprofv3 --kernel-trace --pmc $grp -d $O/${v}_$i -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/r05/step_only.py GINet 40 > $O/${v}_$i.log 2>&1 < /dev/null)
  done
done
python - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
rows = collections.OrderedDict()
for v in "exit1 exit3 exit4 exit5 exit6 exit8 exit9 exit10 exit11 exit12 exit14 full".split():
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(O, v + "_*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "k_step3" not in row["Kernel_Name"]:
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    rows[v] = {k: s / max(n, 1) for k, (s, n) in acc.items()}
names = sorted({k for r in rows.values() for k in r})
print("%-8s " % "variant" + " ".join("%22s" % n for n in names))
for v, r in rows.items():
    print("%-8s " % v + " ".join("%22.0f" % r.get(n, float("nan")) for n in names))
PY
