cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
for net in GINet sGAT FoutNet; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --net $net --no-cpu-baseline > $O/bench_$net.json 2> $O/bench_$net.err
  python -c "
import json,sys;d=json.loads(open('$O/bench_$net.json').read());k=d.get('roofline',{}).get('kernels',{})
print('$net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'epoch', {kk: (round(vv,2) if isinstance(vv, float) else vv) for kk, vv in (d.get('extra',{}).get('epoch_loop',{}) or {}).items() if 'us_per' in kk})"
done
