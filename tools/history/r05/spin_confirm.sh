#!/bin/bash
# confirmation of the scalar-pinned run-time layout in the tree build (base = the build before: variants/libdrgnn_prev.so)
cd $GRAFT_REPO_ROOT
V=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_prev.so
timeout 900 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_width_classes.py tests/test_gpu_parity.py tests/test_gpu_epoch.py -x -q -m gpu < /dev/null 2>&1 | tail -3
O=gpurun_out/spin2; mkdir -p $O
for r in 1 2; do for cfg in "32 64 rebuilt" "32 64 cached" "32 128 rebuilt" "32 128 cached" "48 128 cached" "48 64 cached"; do set -- $cfg; for lib in prev tree; do
  L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; [ $lib = prev ] && L=$V
  DRGNN_LIB=$L timeout 60 python bench.py --net sGAT --topology $3 --n-feat $1 --graphs-per-gpu $2 --no-cpu-baseline --epoch-graphs 0 --no-other-nets \
     --step-layout noclass --min-seconds 1.5 < /dev/null 2>/dev/null > $O/${lib}_$1_$2_$3_$r.json
  python - $O/${lib}_$1_$2_$3_$r.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "%.2f us/step  kernel %.2f" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_us"]))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done; done; done
