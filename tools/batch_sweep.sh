for b in 64 128 256 512 1024 2048; do
  python bench.py --graphs-per-gpu $b --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/sweep_$b.log 2>&1
done
for b in 64 256 1024; do
  python bench.py --graphs-per-gpu $b --steps 100 --warmup 10 --no-cpu-baseline --net sGAT > gpurun_out/sweep_sgat_$b.log 2>&1
  python bench.py --graphs-per-gpu $b --steps 100 --warmup 10 --no-cpu-baseline --net FoutNet > gpurun_out/sweep_fout_$b.log 2>&1
done
echo done
