cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
for r in 1 2; do
for net in sGAT FoutNet; do
for lay in old af1 auto; do
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net $net --step-layout $lay 2>gpurun_out/r04c/err_${net}_$lay.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$lay $net', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])" | tee -a gpurun_out/r04c/ab.txt
done; done; done
