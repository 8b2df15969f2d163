cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tl; mkdir -p $O
COLD_MIX=tiles COLD_PRETOUCH=0 timeout 150 python tools/r05/cold_timeline.py pre0 GINet 32 0 2>&1 | grep "^timeline\|^pretouch\|Error\|error" >> $O/tl5.txt
COLD_MIX=tiles COLD_PRETOUCH=1 timeout 150 python tools/r05/cold_timeline.py pre1 GINet 32 0 2>&1 | grep "^timeline\|^pretouch\|Error\|error" >> $O/tl5.txt
cat $O/tl5.txt
