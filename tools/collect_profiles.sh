#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (separate
# runs, counters never combined with API tracing) of the default bench, plus the bench line itself.
# usage: tools/collect_profiles.sh <tag> [net]   -> gpurun_out/<tag>/{stats,fetch,write,fetch_cached,write_cached,sq1,sq2}/..., benchline.json
#        then here: python tools/summarize_profile.py <name> gpurun_out/<tag> [net]   (writes profiles/<name>_*)
set -e
# (every rocprofv3 run under `timeout` with stdin closed: a profiler waiting on a terminal once cost a whole gpurun call)
TAG=${1:-prof}
NET=${2:-GINet}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python bench.py --net $NET --min-seconds 1 --no-cpu-baseline --epoch-graphs 0 --no-dropin > $OUT/stats.log 2>&1 < /dev/null || true
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o run --output-format csv -- python bench.py --net $NET --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/fetch.log 2>&1 < /dev/null || true
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o run --output-format csv -- python bench.py --net $NET --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/write.log 2>&1 < /dev/null || true
# the same two counters of the launch WITHOUT a co-launched builder (cached topology: what NeuralNet.train runs by default)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch_cached -o run --output-format csv -- python bench.py --net $NET --topology cached --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/fetch_cached.log 2>&1 < /dev/null || true
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write_cached -o run --output-format csv -- python bench.py --net $NET --topology cached --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/write_cached.log 2>&1 < /dev/null || true
# SQ counters (own passes, kernel trace only): MFMA busy, wave-cycle breakdown, LDS conflicts
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq1 -o run --output-format csv -- python bench.py --net $NET --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/sq1.log 2>&1 < /dev/null || true
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $OUT/sq2 -o run --output-format csv -- python bench.py --net $NET --steps 40 --warmup 20 --min-seconds 0 --no-cpu-baseline --epoch-graphs 0 --no-dropin --counter-pass > $OUT/sq2.log 2>&1 < /dev/null || true
# (SKIP_LINES=1: the caller takes the bench lines AFTER summarize_profile.py has put the counter summaries of this build in
# place, so that they carry roofline.traffic / mfma_util)
if [ -n "$SKIP_LINES" ]; then find $OUT -name "*.csv" | head -20; exit 0; fi
timeout 300 python bench.py --net $NET > $OUT/benchline.json 2> $OUT/bench.err
timeout 200 python bench.py --net $NET --steps 20 --warmup 5 --no-cpu-baseline --epoch-graphs 0 > $OUT/benchline_driver_args.json 2>> $OUT/bench.err
find $OUT -name "*.csv" | head -20
tail -c 400 $OUT/benchline.json
