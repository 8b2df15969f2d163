cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 900 python tools/r04/check_step2.py GINet sGAT FoutNet 2>&1 | grep -v "^$" | grep -v "Warn\|detach\|ref_losses" > $O/check.log
cat $O/check.log
for n in GINet sGAT FoutNet; do
for r in 1 2; do
python bench.py --no-cpu-baseline --epoch-graphs 0 --min-seconds 1 --net $n 2>$O/err.txt | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('roofline',{}).get('kernels',{})
print('$n', 'us/step', round(d['ms_per_step']*1000,2), 'kernels', [round(v['avg_us'],2) for v in k.values()], 'loss', d['config']['final_loss'])" | tee -a $O/ab.txt
done
done
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
