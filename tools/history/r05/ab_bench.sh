# same-box A/B of two library builds under the driver's bench arguments: tools/r05/ab_bench.sh <variant.so> [net] [rounds]
cd $GRAFT_REPO_ROOT
V=$1; NET=${2:-GINet}; R=${3:-2}
O=gpurun_out/r05_ab; mkdir -p $O
for r in $(seq 1 $R); do
  for lib in base $V; do
    if [ $lib = base ]; then L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; else L=$PWD/deeprank-gnn_amd/csrc/variants/$lib; fi
    DRGNN_LIB=$L timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --net $NET --no-cpu-baseline --no-other-nets --min-seconds 2 > $O/$lib.$NET.$r.json 2>/dev/null
    echo "== $lib $NET round $r"; python tools/r05/bench_brief.py $O/$lib.$NET.$r.json | grep -v "k_topo (own\|without"
  done
done
