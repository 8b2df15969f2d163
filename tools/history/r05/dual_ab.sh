#!/bin/bash
# GINet one-workgroup step with the two branches side by side (STEP3B_DUAL): parity, then same-box A/B against the build before
cd $GRAFT_REPO_ROOT
V=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_prev.so
timeout 900 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_width_classes.py tests/test_gpu_epoch.py -x -q -m gpu -k "GINet or ginet or epoch" < /dev/null 2>&1 | tail -3
O=gpurun_out/dual; mkdir -p $O
for r in 1 2; do for cfg in "128 rebuilt" "128 cached" "256 rebuilt" "256 cached" "1024 rebuilt" "1024 cached" "96 rebuilt"; do set -- $cfg; for lib in prev tree; do
  L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; [ $lib = prev ] && L=$V
  DRGNN_LIB=$L timeout 60 python bench.py --net GINet --topology $2 --graphs-per-gpu $1 --no-cpu-baseline --epoch-graphs 0 --no-other-nets \
     --min-seconds 1.5 < /dev/null 2>/dev/null > $O/${lib}_$1_$2_$r.json
  python - $O/${lib}_$1_$2_$r.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "%.2f us/step  kernel %.2f  %.3f M graphs/s" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_us"], d["value"]/1e6))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done; done; done
