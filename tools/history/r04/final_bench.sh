cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 300 gpurun_out/final/bench.err
python -c "
import json;d=json.load(open('gpurun_out/final/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline'].get('traffic'),d['distinct_batches'],d['epoch_loop']['us_per_batch'],d['cpu_baseline']['value'])"
