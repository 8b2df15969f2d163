# final evidence of the round in ONE gpurun call per net: tools/collect_profiles.sh, summarised ON the box (the raw rocprofv3
# traces exceed what gpurun copies back), keyed by the kernel sources' hash.  usage: tools/r05/final_collect.sh <net> <name>
cd $GRAFT_REPO_ROOT
NET=$1; NAME=${2:-r05_v1}
mkdir -p gpurun_out/r05f
timeout 1500 bash tools/collect_profiles.sh r05f/$NET $NET > gpurun_out/r05f/collect_$NET.log 2>&1
python tools/summarize_profile.py ${NAME}_$NET gpurun_out/r05f/$NET $NET > gpurun_out/r05f/summarize_$NET.log 2>&1
mkdir -p gpurun_out/r05f/out
cp profiles/${NAME}_${NET}_* gpurun_out/r05f/out/
rm -rf gpurun_out/r05f/$NET
ls gpurun_out/r05f/out | head -20
tail -c 300 gpurun_out/r05f/out/${NAME}_${NET}_benchline_driver_args.json
