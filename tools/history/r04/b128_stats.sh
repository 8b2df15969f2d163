# rocprofv3 kernel-trace stats of the bench at the reference's training batch size (128 per GPU), three nets, rebuilt + cached
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/b128; mkdir -p $O; rm -f $O/b128_kernel_stats.txt
export TMPDIR=/tmp
for net in GINet sGAT FoutNet; do for mode in rebuilt cached; do
  timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/b128_${net}_$mode -o run --output-format csv -- python bench.py --net $net --topology $mode --graphs-per-gpu 128 --min-seconds 1 --no-cpu-baseline --epoch-graphs 0 > $O/${net}_$mode.json 2> $O/${net}_$mode.err
  echo "== $net $mode batch 128: $(python -c "import json;d=json.load(open('$O/${net}_$mode.json'));print('%.2f us per step, %.3f M graphs/s' % (d['ms_per_step']*1000, d['value']/1e6))")" >> $O/b128_kernel_stats.txt
  f=$(find /tmp/b128_${net}_$mode -name "*kernel_stats.csv" | head -1)
  head -4 "$f" | cut -c1-170 >> $O/b128_kernel_stats.txt
done; done
cat $O/b128_kernel_stats.txt
