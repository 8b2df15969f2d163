#!/bin/bash
# capacity class for the 48-wide aggregation-first kernels: parity first, then a same-box A/B (default vs --step-layout noclass)
# at 48 features, batch 64 / 128, rebuilt / cached, three nets
cd $GRAFT_REPO_ROOT
O=gpurun_out/cls48; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_fullsize.py tests/test_gpu_width_classes.py -x -q -m gpu -k "capacity_class or reference_golden or other_widths" < /dev/null 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
for r in 1 2; do for net in GINet sGAT FoutNet; do for B in 64 128; do for mode in rebuilt cached; do for lay in default noclass; do
  timeout 60 python bench.py --net $net --topology $mode --n-feat 48 --graphs-per-gpu $B --no-cpu-baseline --epoch-graphs 0 --no-other-nets \
     --step-layout ${lay/default/auto} --min-seconds 1.5 < /dev/null 2>/dev/null > $O/${net}_${B}_${mode}_${lay}_$r.json
  python - $O/${net}_${B}_${mode}_${lay}_$r.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "%.2f us/step  kernel %.2f" % (d["ms_per_step"]*1e3, d["roofline"]["kernel_us"]))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done; done; done; done; done
