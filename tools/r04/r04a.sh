cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
for net in GINet sGAT FoutNet; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --net $net --no-cpu-baseline --epoch-graphs 0 > gpurun_out/r04a/bench_$net.json 2> gpurun_out/r04a/bench_$net.err
  python -c "
import json,sys;d=json.loads(open('gpurun_out/r04a/bench_$net.json').read());k=d.get('roofline',{}).get('kernels',{})
print('$net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()])"
done
