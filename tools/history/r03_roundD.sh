#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for v in base skip7 exit8 exit9 exit11; do
  DRGNN_LIB=deeprank-gnn_amd/csrc/ablate/libdrgnn_$v.so timeout 120 python tools/time_graph.py $v 2>/dev/null | grep "^graph" | tee -a $OUT/timeline_graph_GINet.txt
done
for net in GINet; do timeout 200 python tools/phase_timing_step.py $net 2>/dev/null | tee $OUT/phase_$net.log | tail -45; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 2 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step']*1000, json.dumps(d['epoch_loop'], indent=1))"
