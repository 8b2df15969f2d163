#!/bin/bash
# side-by-side one-workgroup GINet step at 48 features (branch 1's Z1 aliased): parity, then same-box A/B against the build before
cd $GRAFT_REPO_ROOT
V=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_prev.so
timeout 900 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_width_classes.py tests/test_gpu_epoch.py -x -q -m gpu -k "GINet or ginet or epoch" < /dev/null 2>&1 | tail -3
O=gpurun_out/dual48; mkdir -p $O
for r in 1 2; do for cfg in "48 128 rebuilt" "48 256 rebuilt" "48 256 cached" "48 1024 cached" "32 128 rebuilt" "32 256 cached"; do set -- $cfg; for lib in prev tree; do
  L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; [ $lib = prev ] && L=$V
  DRGNN_LIB=$L timeout 60 python bench.py --net GINet --n-feat $1 --topology $3 --graphs-per-gpu $2 --no-cpu-baseline --epoch-graphs 0 --no-other-nets \
     --min-seconds 1.5 < /dev/null 2>/dev/null > $O/${lib}_$1_$2_$3_$r.json
  python - $O/${lib}_$1_$2_$3_$r.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "%.2f us/step  kernel %.2f  %.3f M graphs/s" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_us"], d["value"]/1e6))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done; done; done
