#!/bin/bash
# sGAT / FoutNet inference launches with the capacity-class layout: parity, then bench.py's inference_loop, prev vs tree, same box
cd $GRAFT_REPO_ROOT
V=$PWD/deeprank-gnn_amd/csrc/variants/libdrgnn_prev.so
timeout 900 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_width_classes.py tests/test_gpu_epoch.py tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_layers.py -x -q -m gpu < /dev/null 2>&1 | tail -3
O=gpurun_out/inf2; mkdir -p $O
for r in 1 2; do for net in sGAT FoutNet; do for lib in prev tree; do
  L=$PWD/deeprank-gnn_amd/csrc/libdrgnn.so; [ $lib = prev ] && L=$V
  DRGNN_LIB=$L timeout 120 python bench.py --net $net --no-cpu-baseline --no-other-nets --min-seconds 0.5 < /dev/null 2>/dev/null | tail -1 > $O/${lib}_${net}_$r.json
  python - $O/${lib}_${net}_$r.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read()); i=d["inference_loop"]
print(sys.argv[1].split('/')[-1], " ".join("%s %.2f us (%.2f M/s)" % (k.replace("batch",""), v["us_per_batch"], v["graphs_per_s"]/1e6) for k,v in i.items() if isinstance(v,dict) and "us_per_batch" in v))
P
done; done; done
